#!/usr/bin/env python3
"""Benchmark of the helmnet IterativeSolver inference loop on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one solver iteration (single_step, reference hybridnet.py:558-584) of one batch
of 32 synthetic 256x256 sound-speed maps per GPU (BASELINE.json configs[1]; ring phantoms
drawn from the reference's training distribution, shipped-checkpoint weights, fp32).  Inputs
are resident in HBM before the timed region.  Batches shard across GPUs with no data-path
collective (weak scaling); the only RCCL call is the tiny residual-norm all-reduce.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 (vector == f32-input MFMA), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def kernel_macs(n: int, depth: int = 4) -> dict:
    """Algorithmic multiply-accumulates per SAMPLE per launch of each kernel."""
    def dc(cin, cm, co, px):
        return px * 9 * (cin * cm + cm * co)
    m = {"inc": dc(6, 8, 8, n * n)}
    for d in range(depth):
        nd = n >> d
        m[f"conv_signal{d}"] = dc(10, 8, 8, nd * nd)
        m[f"conv_state{d}"] = dc(10, 2, 2, nd * nd)
        m[f"down{d}"] = 64 * 8 * 8 * (nd // 2) ** 2
        m[f"up{d}"] = 16 * 8 * 8 * nd * nd
        m[f"decode{d}"] = dc(16, 8, 8, nd * nd) + (16 * nd * nd if d == 0 else 0)
    m["bottleneck"] = dc(8, 8, 8, (n >> depth) ** 2)
    return m


# rocprofv3 kernel names of the kernels that are launched once per step (the per-level kernels share
# one name across levels); used to look the dominant kernel's measured HBM traffic up in
# profiles/*_traffic.json (FETCH_SIZE + WRITE_SIZE of the same command, tools/summarize_profiles.py)
ROCPROF_NAME = {"decode0": "k_dc_mfma_s<8, 8, 0, 1>", "inc": "k_dc_mfma_s<2, 2, 2, 0>",
                "spectral_rows": "k_spec_rows<256>", "spectral_cols": "k_spec_cols<256, 16>"}


def measured_traffic(kernel: str, n: int, batch: int):
    """HBM bytes per launch from the committed PMC summary (valid for the default 256^2 x 32 workload only)."""
    if n != 256 or batch != 32 or kernel not in ROCPROF_NAME:
        return None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        with open(path) as f:
            t = json.load(f)
        if ROCPROF_NAME[kernel] in t:
            return t[ROCPROF_NAME[kernel]]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def spectral_bytes(n: int) -> int:
    """Compulsory HBM bytes per sample of the residual: read wf(2)+k_sq(1), write res(2) planes."""
    return 5 * 4 * n * n


def cpu_baseline(sos_cpu, n, loc, budget_s=20.0, max_iters=8):
    """The CPU oracle (stock PyTorch CPU ops, oneDNN/MKL) timed on the host cores: a bounded
    sample of the SAME workload (same batch, same maps), iterations/s of the batch."""
    from oracle import helmnet_oracle as O
    w = {}
    with np.load(os.path.join(ROOT, "tests", "golden", "jcp_weights.npz")) as z:
        w = {k: torch.from_numpy(z[k]) for k in z.files}
    t = O.SpectralTables(n, 8, 2, 1.0)
    src = O.point_source_map(n, loc, 10.0)
    with torch.no_grad():
        k_sq, wf = O.get_initials(sos_cpu, 1.0)
        st = [torch.zeros(sos_cpu.shape[0], 2, s, s) for s in O.state_dims(n, 4)]
        res = O.get_residual(wf, k_sq, src, t)
        wf, res, st = O.single_step(wf, k_sq, res, st, w, src, t)  # warm-up
        t0 = time.perf_counter()
        it = 0
        while it < max_iters and (time.perf_counter() - t0) < budget_s:
            wf, res, st = O.single_step(wf, k_sq, res, st, w, src, t)
            it += 1
        dt = time.perf_counter() - t0
    return it / dt, it, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="SoS maps per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--lanes", type=int, default=1, help="hn_step pipeline lanes (sub-batches on parallel streams); "
                    "2 gives about +5 %% it/s but kernels of the two lanes overlap, so per-kernel timings (roofline) blur")
    ap.add_argument("--unet-impl", default=None, choices=["valu", "bf16x3", "bf16x2", "fp16"],
                    help="experiments only (sets HN_UNET_IMPL): 'bf16x3' = split-bf16 DoubleConv kernels with fp32-accurate "
                         "products; the default and the reported metric is the fp32 matrix-core path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel time table (stderr)")
    args = ap.parse_args()

    os.environ["HN_STREAMS"] = str(args.lanes)   # read by libhelmnet_hip.so when its first hn_step runs
    if args.unet_impl:
        os.environ["HN_UNET_IMPL"] = args.unet_impl
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torchrun: same code path for every N
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from helmnet_amd import IterativeSolver
    from helmnet_amd.distributed import allreduce_residual_norms
    from helmnet_amd.phantoms import readme_sos, ring_sos_batch

    n, B, K, W = args.size, args.batch, args.steps, args.warmup
    loc = [30, n // 2]
    solver = IterativeSolver.from_exported_weights()
    solver.freeze()
    solver.to(dev)
    solver.set_domain_size(n, source_location=loc)
    sos_np = ring_sos_batch(n, B, seed=rank)          # each rank solves its own shard of maps
    if rank == 0 and n == 256:
        sos_np[0] = readme_sos()[0]                   # README example as sample 0 (SURVEY 8d cfg 2)
    sos = torch.from_numpy(sos_np).to(dev)
    eng = solver.engine()
    eng.reserve(B)

    k_sq, wf = solver.get_initials(sos)
    solver.f.clear_states(wf)
    res = solver.get_residual(wf, k_sq)
    st = solver.f.get_states(flatten=True).contiguous()
    k_sq, src = k_sq.contiguous(), solver.source.detach().contiguous()
    rmse = torch.zeros(max(K, W, 1), B, device=dev)

    # warm-up: W untimed steps, with every kernel bracketed by events to find the dominant one
    eng.profile_enable(None)
    if W > 0:
        eng.step(wf, res, st, k_sq, src, W, rmse_hist=rmse[:W])
    torch.cuda.synchronize()
    prof = eng.profile_collect()
    pmin = eng.profile_min()    # shortest launch per kernel: the host cannot keep up with ~70 API calls per step
    dominant = max(pmin, key=pmin.get) if pmin else "decode0"
    dom_id = [i for i in range(eng.KERNEL_IDS) if eng.kernel_name(i) == dominant][0]
    eng.profile_enable([dom_id])
    stride = max(1, K // 32)                          # ~32 sampled launches: event pairs cost ~6 us of stream gap each
    eng.profile_stride(stride)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    eng.step(wf, res, st, k_sq, src, K, rmse_hist=rmse[:K])
    worst = allreduce_residual_norms(rmse[K - 1], op="max")   # the path's only collective (<= 128 B)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    dom_ms, dom_cnt = eng.profile_collect().get(dominant, (0.0, 0))
    eng.profile_enable([])
    eng.profile_stride(1)

    if rank == 0:
        macs = kernel_macs(n)
        total_flops = 2.0 * sum(macs.values()) * B
        # hn_step may split the batch over pipeline lanes: flops per LAUNCH = flops per step / launches per step
        per_step = max(1, round(dom_cnt * stride / max(1, K)))
        dom_flops = 2.0 * macs[dominant] * B / per_step if dominant in macs else 0.0
        roof = None
        traffic, traffic_src = measured_traffic(dominant, n, B)
        if dom_cnt:
            avg_s = dom_ms / dom_cnt * 1e-3
            if dominant in macs:
                ach = dom_flops / avg_s / 1e12
                roof = {"kernel": dominant, "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                        "traffic_source": traffic_src,
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches": dom_cnt,
                        "samples_per_launch": B // per_step, "flops_per_launch": dom_flops}
            else:
                byts = spectral_bytes(n) * B / per_step
                ach = byts / avg_s / 1e9
                roof = {"kernel": dominant, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": traffic_src,
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches": dom_cnt, "bytes_per_launch": byts}
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (torchrun pins OMP threads to 1 per rank)
            its, cnt, secs = cpu_baseline(torch.from_numpy(sos_np), n, loc)
            cpu = {"value": round(its, 4), "unit": "iterations/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{cnt} single_step iterations of the same {B}x{n}x{n} batch ({secs:.1f} s), "
                             "oracle/helmnet_oracle.py on PyTorch CPU ops"}
        final_rmse = rmse[K - 1].float().cpu().numpy()
        line = {
            "metric": f"solver iterations/sec (whole node), {n}^2 domain batch={B}",
            "value": round(world * K / dt, 2),
            "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}x{n} ring-phantom SoS maps, batch={B} per GPU, point source {loc}, "
                                   "shipped jcp checkpoint weights, fp32 (BASELINE configs[1])",
                       "batch_per_gpu": B, "domain": n, "lanes": args.lanes, "unet_impl": args.unet_impl or "fp32-mfma", "parallelism": f"dp{world} (batch shards, no data-path collective)"},
            "sample_iterations_per_s": round(world * B * K / dt, 1),
            "unet_tflops": round(total_flops * K / dt / 1e12, 2),
            "residual_rmse_after_timed_steps": {"median": float(np.median(final_rmse)), "max": float(worst.max().item())},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        # secondary line for the HBM-bound part of the path (north_star: "achieved HBM GB/s for the FFT path"):
        # compulsory bytes of get_residual (5 planes per sample) over the shortest bracketed launches of the two
        # spectral kernels in the fully bracketed warm-up pass
        sp_us = (pmin.get("spectral_cols", 0.0) + pmin.get("spectral_rows", 0.0)) * 1e3
        if sp_us > 0:
            sp_bytes = spectral_bytes(n) * B
            line["hbm_path"] = {"kernels": ["spectral_cols", "spectral_rows"], "bound": "hbm", "bytes_per_step": sp_bytes,
                                "us_per_step": round(sp_us, 2), "achieved": round(sp_bytes / (sp_us * 1e-6) / 1e9, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sp_bytes / (sp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if args.breakdown:
            # event brackets around EVERY kernel stall the stream (the host cannot issue ~70 API calls per
            # step fast enough), so the average over-states a kernel that follows a host gap; the shortest
            # launch is the robust per-kernel figure and is what the dominant-kernel pick uses
            tot = sum(pmin.values()) * 1e3
            for k, ms_min in sorted(pmin.items(), key=lambda kv: -kv[1]):
                us = ms_min * 1e3
                fl = 2.0 * macs.get(k, 0) * B
                ms, c = prof.get(k, (0.0, 1))
                print(f"  {k:14s} min {us:8.1f} us/launch  {100 * us / tot:5.1f} %  "
                      f"{fl / (us * 1e-6) / 1e12 if fl else 0:7.1f} TFLOP/s   (bracketed avg {ms / max(1, c) * 1e3:8.1f} us)", file=sys.stderr)
            print(f"  sum of per-kernel minima {tot:.1f} us/step (warm-up pass, all kernels bracketed; "
                  f"conv_state* run on the side stream, overlapped)", file=sys.stderr)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
