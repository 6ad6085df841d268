"""(Fixed in r5 by zero_async: the numbers below are the library BEFORE it.)  A captured training step (torch.cuda.graph around hn_train_grad) replays garbage after ANY library call on the legacy default stream.

    python tools/graph_null_stream_probe.py N B [side|null] [refuse|norefuse] [null|side2|side|fwd|resid|resid_side2|rmse|rmse_side2|torch|none]

Sequence: eager step on a created stream, capture on it, two replays (equal to the eager gradient), then ONE intervening call (5th argument), then a replay.
[seen, r5, 64 4 side norefuse ...] null (eager hn_train_grad on the default stream), fwd (solver.forward), resid (hn_residual): replay garbage / NaN in 3 of 4 runs;
side2 / side (eager step on another / the same created stream), torch (torch kernels on the default stream), none: replay exact.  The round-4 library behaves the same.
rmse (hn_rmse on the default stream: one hipMemsetAsync and two scratch-free kernels) breaks the replay too; rmse_side2 / resid_side2 (the same calls on a created stream) do not:
it is not a particular kernel or its scratch memory.  hipmemset (the HIP runtime's hipMemsetAsync on the default stream, no library of ours involved) breaks it as well: the
graph's memset nodes are the victims.  With zero_async (no memset nodes in the graph, no hipMemsetAsync in the eager calls) every variant replays exactly.
"""
import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from helmnet_amd import IterativeSolver
from helmnet_amd.engine import pack_weights
from helmnet_amd.phantoms import ring_sos_batch
n, b = int(sys.argv[1]), int(sys.argv[2])
with np.load("/root/repo/tests/golden/jcp_weights.npz") as z: weights = {k: torch.from_numpy(z[k]) for k in z.files}
s = IterativeSolver.from_exported_weights(); s.to("cuda:0")
s.set_domain_size(n, source_location=[n - 14, n // 2])
eng = s.engine()
sos = torch.from_numpy(ring_sos_batch(n, b, seed=8)).cuda()
out = s.forward(sos, num_iterations=3, return_wavefields=True, return_states=True)
args = [out["wavefields"][-1].contiguous(), out["residuals"][-1].contiguous(), out["states"][-1].contiguous(),
        ((1.0 / sos) ** 2).contiguous(), s.source.detach().repeat(b, 1, 1, 1).contiguous()]
blob = torch.from_numpy(pack_weights(weights)).cuda()
g = torch.zeros_like(blob)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eager = eng.train_grad(blob, *args, 3, 1e4, grad=g)
    torch.cuda.synchronize()
    want = g.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        captured = eng.train_grad(blob, *args, 3, 1e4, grad=g)
    for i in range(2):
        g.zero_(); captured["loss"].zero_(); graph.replay(); torch.cuda.synchronize()
        print("replay", i, bool(torch.equal(g, want)), float((g - want).abs().max()))
how = sys.argv[5] if len(sys.argv) > 5 else "null"
if how == "null":
    again = eng.train_grad(blob, *args, 3, 1e4, grad=g)
elif how == "side2":
    with torch.cuda.stream(torch.cuda.Stream()):
        again = eng.train_grad(blob, *args, 3, 1e4, grad=g)
elif how == "side":
    with torch.cuda.stream(side):
        again = eng.train_grad(blob, *args, 3, 1e4, grad=g)
elif how == "fwd":
    out2 = s.forward(sos, num_iterations=3)
elif how == "torch":
    x = torch.randn(1 << 22, device="cuda"); y = (x * 2).sum(); z = torch.fft.fft(x)
elif how == "resid":
    r = eng.residual(args[0], args[3][:, :1] if args[3].dim() == 4 else args[3], s.source.detach().contiguous())
elif how == "resid_side2":
    with torch.cuda.stream(torch.cuda.Stream()):
        r = eng.residual(args[0], args[3], s.source.detach().contiguous())
elif how == "outconv":
    xx = torch.randn(2, 8, 64, 64, device="cuda"); r = eng.lib and None
elif how == "rmse":
    r = eng.rmse(args[1])
elif how == "rmse_side2":
    with torch.cuda.stream(torch.cuda.Stream()):
        r = eng.rmse(args[1])
elif how == "hipmemset":     # no library of ours involved: the HIP runtime called directly on the legacy default stream
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    x = torch.ones(1024, device="cuda")
    print("hipMemsetAsync(default stream) ->", hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, 4096, ctypes.c_void_p(0)))
elif how == "none":
    pass
torch.cuda.synchronize()
print("again", how, bool(torch.equal(g, want)), float((g - want).abs().max()))
if len(sys.argv) <= 4:
    try:
        eng.train_grad(blob, *args, 25, 1e4, grad=g)
    except Exception as e:
        print("refused")
mode = sys.argv[3] if len(sys.argv) > 3 else "null"
if mode == "side":
    with torch.cuda.stream(side):
        graph.replay(); torch.cuda.synchronize()
else:
    graph.replay(); torch.cuda.synchronize()
print("replay after on", mode, bool(torch.equal(g, want)), float((g - want).abs().max()))
