"""CPU oracle for the retrieval row -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Restates ``make_prediction`` (reference scripts/inference_and_eval.py:414-445).  The search itself lives in two third-party
dependencies that are not vendored under /root/reference: scikit-learn ``preprocessing.normalize(norm="l2", axis=1)`` (rows
divided by their L2 norm, zero rows left as they are) and faiss ``IndexFlatIP`` (pinned only as "faiss" in the reference's
requirements; exhaustive inner-product search, results per query sorted by decreasing score).  scikit-learn is importable
here and ``topk_ip`` is checked against its ``normalize`` in tests/test_01_oracle_golden.py; faiss is absent, so the search half
is "parity unpinned" against faiss itself and anchored on its published definition (brute-force float32 inner products).
The accuracy helpers are pinned against the reference's own functions through tests/golden/retrieval.json
(oracle/gen_golden.py).
"""
import numpy as np


def l2_normalize_rows(x):
    """sklearn.preprocessing.normalize(x, norm='l2', axis=1).astype(float32)  (reference :416-417)."""
    x = np.asarray(x, dtype=np.float64)
    n = np.sqrt((x * x).sum(axis=1, keepdims=True))
    n[n == 0.0] = 1.0
    return (x / n).astype(np.float32)


def topk_ip(query_feature, keys_feature, max_k):
    """faiss IndexFlatIP(d).add(keys).search(queries, max_k) on normalised rows (reference :415-422).

    Returns (similarities f32 [Q, k], indices int64 [Q, k]); ties resolved towards the lower key index.
    """
    q = l2_normalize_rows(query_feature)
    k = l2_normalize_rows(keys_feature)
    scores = q.astype(np.float64) @ k.astype(np.float64).T
    order = np.lexsort((np.broadcast_to(np.arange(scores.shape[1]), scores.shape), -scores), axis=1)[:, :max_k]
    return np.take_along_axis(scores, order, axis=1).astype(np.float32), order.astype(np.int64)


LEVELS = ["order", "family", "genus", "species"]


def make_prediction(query_feature, keys_feature, keys_label, max_k=5):
    _, indices = topk_ip(query_feature, keys_feature, max_k)
    return [{level: [keys_label[i][level] for i in row] for level in LEVELS} for row in indices]


def top_k_micro_accuracy(pred_list, gt_list, k_list):
    """reference :448-464"""
    out = {}
    for k in k_list:
        out[k] = {}
        for level in LEVELS:
            ok = 0
            for p, g in zip(pred_list, gt_list):
                ok += g[level] in p[level][:k]
            out[k][level] = ok * 1.0 / len(pred_list)
    return out


def top_k_macro_accuracy(pred_list, gt_list, k_list):
    """reference :467-511"""
    macro, per_class = {}, {}
    for k in k_list:
        macro[k], per_class[k] = {}, {}
        for level in LEVELS:
            n, ok = {}, {}
            for p, g in zip(pred_list, gt_list):
                n[g[level]] = n.get(g[level], 0) + 1
                ok[g[level]] = ok.get(g[level], 0) + (g[level] in p[level][:k])
            per_class[k][level] = {c: ok[c] * 1.0 / n[c] for c in n}
            s = 0
            for c in n:
                s = s + ok[c] * 1.0 / n[c]
            macro[k][level] = s / len(n)
    return macro, per_class


def retrieval_case(seed=5, n_keys=60, n_query=40, max_k=5):
    """Seeded taxonomy labels + random key indices per query: the inputs of the accuracy helpers."""
    rng = np.random.RandomState(seed)

    def label(i):
        sp = int(i)
        return {"order": f"o{sp % 3}", "family": f"f{sp % 5}", "genus": f"g{sp % 7}", "species": f"s{sp % 11}"}

    keys_label = [label(rng.randint(0, 1000)) for _ in range(n_keys)]
    gt_list = [label(rng.randint(0, 1000)) for _ in range(n_query)]
    indices = rng.randint(0, n_keys, size=(n_query, max_k))
    pred_list = [{lv: [keys_label[i][lv] for i in row] for lv in LEVELS} for row in indices]
    return keys_label, gt_list, indices, pred_list
