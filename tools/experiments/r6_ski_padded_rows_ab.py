"""VERDICT r5 #4 — one bounded attempt: would rows of V padded to 64 bytes (T = 11 -> 16 floats, every gathered row one aligned
sector) speed the cell-sorted SKI scatter enough to pay for writing the padded copy?  Needs the library built with
tools/experiments/r6_ski_padded_rows.patch (the scatter reads V with a row stride of RPGP_EXP_SKI_LDV floats).  Measures, at the
C5 shape (N = 391 386, J = 3, G = 1024, T = 11, Gaussian coordinates in locality order as training stores them):
  scatter with 44-byte rows / with 64-byte rows (same values: the histograms must agree bit for bit), alternating, HIP events;
  the cost of producing the padded copy: a strided copy of the N x 11 block into an N x 16 buffer (what executor pass C would add).
Prints one JSON line."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops, _lib
from rpgp_amd.training import locality_order
dev = torch.device("cuda:0")
N, J, G, T = 391386, 3, 1024, 11
g = torch.Generator().manual_seed(5)
X = torch.randn(N, J, generator=g)
Q = torch.linalg.qr(torch.randn(J, J, generator=g))[0]
Z = (X @ Q).contiguous()
Z = Z[locality_order(Z)].contiguous().to(dev)
V = torch.randn(N, T, generator=g).to(dev)
Vp = torch.zeros(N, 16, device=dev)
Vp[:, :T] = V
gp = ops.ski_grid(Z, None, G)
plan = ops.SkiPlan(Z, gp, G)
lib = _lib.load()
ws = torch.empty(int(lib.rpgp_ski_workspace_bytes(J, G, T)), dtype=torch.uint8, device=dev)
hist = {k: torch.empty(J, G, T, dtype=torch.float64, device=dev) for k in ("plain", "padded")}
st = torch.cuda.current_stream().cuda_stream


def scatter(kind):
    if kind == "padded":
        os.environ["RPGP_EXP_SKI_LDV"] = "16"
        src = Vp
    else:
        os.environ.pop("RPGP_EXP_SKI_LDV", None)
        src = V
    rc = lib.rpgp_ski_scatter_planned(plan.buf.data_ptr(), src.data_ptr(), hist[kind].data_ptr(), N, J, G, T, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, rc


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {"plain_us": [], "padded_us": [], "pad_copy_us": []}
for _ in range(4):
    res["plain_us"].append(round(timed(lambda: scatter("plain")), 2))
    res["padded_us"].append(round(timed(lambda: scatter("padded")), 2))
    res["pad_copy_us"].append(round(timed(lambda: Vp[:, :T].copy_(V)), 2))
res["bitwise_equal"] = bool(torch.equal(hist["plain"], hist["padded"]))
res["what"] = ("C5 shape, T = 11: rpgp_ski_scatter_planned (cell scatter + cell sums) with 44-byte rows of V against rows padded to "
               "64 bytes; pad_copy = the strided copy that would produce the padded block (a lower bound of what pass C would add)")
print(json.dumps(res))
