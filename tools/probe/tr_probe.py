"""Empirical check of ds_read_b64_tr_b16 semantics (cdna_hip_programming.md T10) before relying on it."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tr_probe.so"))
inp = torch.arange(64 * 64, dtype=torch.int16, device="cuda")
out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
lib.run_probe(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().reshape(64, 4)
ok = True
for l in range(64):
    g, i = l >> 4, l & 15
    want = [(4 * g + q) * 64 + i for q in range(4)]   # column i of rows 4g..4g+3
    if o[l].tolist() != want:
        ok = False
        print("lane", l, "got", o[l].tolist(), "want", want)
print("tr_b16 semantics as documented:", ok)
