"""bioscanclip package of the MI355X-native BIOSCAN-CLIP hot path."""
