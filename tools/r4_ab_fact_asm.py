"""A/B of the headline launch in ONE process on one box (boxes differ by ~4 %): hand-scheduled asm loop (RPGP_FACT_ASM=1)
against the compiler-scheduled mvm_fact_kernel<20,1,2> (=0), alternating, HIP-event kernel times (rpgp_profile_*)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def run(flag, reps=20, taper="1"):
    os.environ["RPGP_FACT_ASM"] = flag
    os.environ["RPGP_TAPER"] = taper
    for _ in range(3):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
    torch.cuda.synchronize()
    lib.rpgp_profile_begin()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
    lib.rpgp_profile_end(ctypes.byref(ms), ctypes.byref(cnt))
    return round(ms.value, 4), round(e0.elapsed_time(e1) / reps, 4)


res = {"N": N, "pairs": []}
for _ in range(4):
    a = run("1")
    u = run("1", taper="0")
    c = run("0")
    res["pairs"].append({"asm_tapered_kernel_ms": a[0], "asm_tapered_step_ms": a[1], "asm_uniform_kernel_ms": u[0],
                         "compiler_kernel_ms": c[0], "compiler_step_ms": c[1]})
o1 = None
os.environ["RPGP_TAPER"] = "1"
os.environ["RPGP_FACT_ASM"] = "1"; oa = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1).clone()
os.environ["RPGP_FACT_ASM"] = "0"; oc = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1).clone()
res["rel_diff_asm_vs_compiler"] = float((oa - oc).norm() / oc.norm())
print(json.dumps(res))
