"""Inference throughput with the default stream vs a torch side stream as the CALLER's stream (does the library's side stream share a hardware queue with it?).
Usage: python tools/bench_caller_stream.py [--steps 300]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import bench
    from helmnet_amd import IterativeSolver
    dev = torch.device("cuda:0")
    s = IterativeSolver.from_exported_weights()
    s.to(dev)
    eng, _, (wf, res, st, k_sq, src) = bench.make_problem(s, 256, 32, [30, 128], 0, dev, True)
    out = {}
    streams = {"default": torch.cuda.default_stream(dev)}
    for i in range(3):
        streams[f"torch side stream {i}"] = torch.cuda.Stream(device=dev)
    for name, stream in streams.items():
        with torch.cuda.stream(stream):
            eng.step(wf, res, st, k_sq, src, 30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.step(wf, res, st, k_sq, src, a.steps)
            torch.cuda.synchronize()
            out[name] = round(a.steps / (time.perf_counter() - t0), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
