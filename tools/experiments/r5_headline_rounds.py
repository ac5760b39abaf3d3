"""EXPERIMENT: the headline launch (hand-scheduled fused MVM, C4) with exactly-filled rounds of resident workgroups
(RPGP_FUSED_ROUNDS = k: the smallest chunk whose workgroups fit k x 768 slots, no taper) against the tapered plan; one
process, alternating, HIP-event kernel times."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rpgp_amd import ops, _lib
dev = torch.device("cuda:0")
N, d, J = int(os.environ.get("AB_N", 50000)), 20, 20
X, P, ls, V = bench.make_inputs(N, d, J, 1, dev)
Z = ops.project(X, (P / ls[:, None]).contiguous())
prep = ops.Prepared(Z)
lib = _lib.load()
out = torch.empty_like(V)
def run(rounds, reps=20):
    if rounds is None: os.environ.pop("RPGP_FUSED_ROUNDS", None)
    else: os.environ["RPGP_FUSED_ROUNDS"] = str(rounds)
    for _ in range(3):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
    torch.cuda.synchronize()
    lib.rpgp_profile_begin()
    for _ in range(reps):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
    lib.rpgp_profile_end(ctypes.byref(ms), ctypes.byref(cnt))
    return round(ms.value, 4)
res = {"N": N}
ref = None
for rep in range(4):
    for k in (None, 2, 3, 4, 5, 6):
        t = run(k)
        key = "tapered_ms" if k is None else "rounds%d_ms" % k
        res[key] = min(t, res.get(key, 1e9))
        if ref is None: ref = out.clone()
        res["max_rel_diff"] = max(res.get("max_rel_diff", 0.0), float((out - ref).norm() / ref.norm()))
print(json.dumps(res))
