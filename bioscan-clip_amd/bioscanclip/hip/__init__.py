"""HIP compute path (ctypes binding, tensor-level op wrappers, encoder engines).  No torch arithmetic lives here."""
