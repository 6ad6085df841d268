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

# MI355X_MICROARCH.md: dense fp32 (vector == f32-input MFMA) 157.3 TF, bf16 / fp16 MFMA ~2.5 PF, HBM3E 8 TB/s
PEAK_TFLOPS = {"fp32": 157.3, "valu": 157.3, "bf16x3": 2500.0, "bf16x2": 2500.0, "fp16": 2500.0}
# arithmetic type of the UNet products; the bf16 split modes are emulations: N product terms per fp32 product
DTYPE = {"fp32": "f32", "valu": "f32", "bf16x3": "bf16x3 (3-term split emulation of f32, f32 accumulate)",
         "bf16x2": "bf16x2 (2-term split, f32 accumulate)", "fp16": "f16 (f32 accumulate; spectral residual f32)"}
TERMS = {"fp32": 1, "valu": 1, "bf16x3": 6, "bf16x2": 3, "fp16": 1}
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
    m["inc_conv_signal0"] = m["inc"] + m["conv_signal0"]   # the two as ONE launch (HN_OPT_DC_PAIR, hn_dca.hip: k_dc_asm_pair); never summed with its parts
    return m


def kernel_macs_executed(n: int, name: str):
    """MACs the vector-pipe DoubleConv kernels of level 0 actually execute per SAMPLE per launch (hn_dca.hip / hn_dcv.hip: 16 x 64 output tiles,
    conv1 on the 18 x 66 mid tile incl. its halo, decode0's conv2 composed with the out-conv into a 2-channel 3x3), or None for other kernels."""
    if name == "inc_conv_signal0":
        return kernel_macs_executed(n, "inc") + kernel_macs_executed(n, "conv_signal0")
    cin = {"inc": 6, "conv_signal0": 10, "decode0": 16}.get(name)
    if cin is None:
        return None
    tiles = -(-n // 64) * -(-n // 16)
    return tiles * (18 * 66 * 8 * cin * 9 + 16 * 64 * (2 if name == "decode0" else 8) * 8 * 9)


def unet_macs(n: int) -> int:
    """MACs of one HybridNet evaluation per sample (every reference layer once)."""
    return sum(v for k, v in kernel_macs(n).items() if k != "inc_conv_signal0")


def kernel_bytes(n: int, depth: int = 4) -> dict:
    """Compulsory HBM bytes per SAMPLE per launch (every input / output plane crosses HBM once)."""
    b = {"inc": 4 * n * n * (6 + 8)}
    for d in range(depth):
        px = (n >> d) ** 2
        b[f"conv_signal{d}"] = 4 * px * (10 + 8)
        b[f"conv_state{d}"] = 4 * px * (10 + 2)
        b[f"down{d}"] = 4 * px * 8 + 4 * (px // 4) * 8
        b[f"up{d}"] = 4 * (px // 4) * 8 + 4 * px * 8
        b[f"decode{d}"] = 4 * px * (16 + (4 if d == 0 else 8))   # d = 0: read-modify-write of the wavefield, no 8-ch output
    b["bottleneck"] = 4 * ((n >> depth) ** 2) * 16
    b["inc_conv_signal0"] = b["inc"] + b["conv_signal0"]
    return b


# rocprofv3 kernel names of the kernels that are launched once per step; used to look the dominant kernel's measured
# HBM traffic up in profiles/*_traffic.json (FETCH_SIZE + WRITE_SIZE of the same command, tools/summarize_profiles.py)
ROCPROF_NAME = {"decode0": ("k_dc_asm<8, 8, 0, 1>", "k_dc_valu<8, 8, 0, 1, false, false>", "k_dc_valu<8, 8, 0, 1, false>", "k_dc_mfma_s<8, 8, 0, 1, false>", "k_dc_mfma_s<8, 8, 0, 1>"),
                "inc": ("k_dc_asm<2, 2, 2, 0>", "k_dc_valu<2, 2, 2, 0, false, false>", "k_dc_valu<2, 2, 2, 0, false>", "k_dc_mfma_s<2, 2, 2, 0, false>", "k_dc_mfma_s<2, 2, 2, 0>"),
                "inc_conv_signal0": ("k_dc_asm_pair",),
                "conv_signal0": ("k_dc_asm<8, 2, 0, 0>", "k_dc_valu<8, 2, 0, 0, false, false>", "k_dc_mfma_s<8, 2, 0, 0, false>"),
                "spectral_rows": ("k_spec8_rows", "k_spec512_rows", "k_spec_rows<256>"),
                "spectral_cols": ("k_spec16_cols_t<16>", "k_spec512_cols_t", "k_spec16_cols_t<32>", "k_spec16_cols", "k_spec_cols<256, 16>"),
                "deep": ("k_deepx<64, 2, false>", "k_deepx<64, 1, false>", "k_deep32<false>")}


def measured_traffic(kernel: str, n: int, batch: int, precision: str):
    """HBM bytes per launch from the committed PMC summary of THIS workload: profiles/r*_traffic.json for the default fp32 256^2 x 32, profiles/r*_512_traffic.json
    for BASELINE configs[3] (512^2 x 16); None for anything else."""
    if (n, batch) not in ((256, 32), (512, 16)) or precision != "fp32" or kernel not in ROCPROF_NAME:
        return None, None
    import glob
    import re
    pat = re.compile(r"r\d+_traffic\.json$" if n == 256 else r"r\d+_512_traffic\.json$")
    for path in sorted((p for p in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")) if pat.search(os.path.basename(p))), reverse=True):
        with open(path) as f:
            t = json.load(f)
        for name in ROCPROF_NAME[kernel]:   # the newest summary that knows one of the kernel's names (newest name first)
            if name in t:
                return t[name]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def spectral_bytes(n: int) -> int:
    """Compulsory HBM bytes per sample of the residual: read wf(2)+k_sq(1), write res(2) planes."""
    return 5 * 4 * n * n


def cpu_baseline(sos_cpu, n, loc, budget_s=14.0):
    """The CPU oracle (stock PyTorch CPU ops, oneDNN/MKL) on the host cores: a bounded sample of the SAME workload.
    BASELINE.md section 3: thread count chosen among {8, 32, 64, physical cores} by a one-iteration probe each; the
    batch as benchmarked (B maps) and B = 1."""
    from oracle import helmnet_oracle as O
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    with np.load(os.path.join(ROOT, "tests", "golden", "jcp_weights.npz")) as z:
        w = {k: torch.from_numpy(z[k]) for k in z.files}
    t = O.SpectralTables(n, 8, 2, 1.0)
    src = O.point_source_map(n, loc, 10.0)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass

    def state(sos):
        k_sq, wf = O.get_initials(sos, 1.0)
        st = [torch.zeros(sos.shape[0], 2, s, s) for s in O.state_dims(n, 4)]
        return wf, k_sq, O.get_residual(wf, k_sq, src, t), st

    def timed(sos, iters_max, budget):
        wf, k_sq, res, st = state(sos)
        wf, res, st = O.single_step(wf, k_sq, res, st, w, src, t)   # warm-up
        t0, it = time.perf_counter(), 0
        while it < iters_max and (time.perf_counter() - t0) < budget:
            wf, res, st = O.single_step(wf, k_sq, res, st, w, src, t)
            it += 1
        return it / (time.perf_counter() - t0), it, time.perf_counter() - t0

    keep = torch.get_num_threads()
    out = {}
    with torch.no_grad():
        probe = {}
        for th in sorted({c for c in (8, 32, 64, phys) if c and c <= (os.cpu_count() or c)}):
            torch.set_num_threads(th)
            probe[th] = timed(sos_cpu, 1, 1e9)[0]
        best = max(probe, key=probe.get)
        torch.set_num_threads(best)
        its, cnt, secs = timed(sos_cpu, 12, budget_s)
        its1, cnt1, secs1 = timed(sos_cpu[:1], 60, 3.0)
    torch.set_num_threads(keep)
    B = sos_cpu.shape[0]
    out = {"value": round(its, 4), "unit": "iterations/s", "cores": best, "kind": "port",
           "sample": f"{cnt} single_step iterations of the same {B}x{n}x{n} batch ({secs:.1f} s), "
                     f"oracle/helmnet_oracle.py on PyTorch CPU ops, {best} threads",
           "cpu_model": model, "physical_cores": phys,
           "thread_probe_it_per_s": {str(k): round(v, 4) for k, v in probe.items()},
           "batch1": {"value": round(its1, 3), "unit": "iterations/s", "sample": f"{cnt1} iterations of sample 0 alone ({secs1:.1f} s)"}}
    return out


def make_problem(solver, n, B, loc, seed, dev, readme_first):
    from helmnet_amd.phantoms import readme_sos, ring_sos_batch
    solver.set_domain_size(n, source_location=loc)
    sos_np = ring_sos_batch(n, B, seed=seed)
    if readme_first and n == 256:
        sos_np[0] = readme_sos()[0]                   # README example as sample 0 (SURVEY 8d cfg 2)
    sos = torch.from_numpy(sos_np).to(dev)
    eng = solver.engine()
    eng.reserve(B)
    k_sq, wf = solver.get_initials(sos)
    solver.f.clear_states(wf)
    res = solver.get_residual(wf, k_sq)
    st = solver.f.get_states(flatten=True).contiguous()
    return eng, sos_np, (wf, res, st, k_sq.contiguous(), solver.source.detach().contiguous())


class Hwmon:
    """Shader clock and board power of the GPU torch runs on (sysfs hwmon, ~4 ms sampling in a thread) while a region runs.  The solver loop does NOT run at
    the nominal 2.4 GHz the `peak` figures assume: the level-0 kernels draw 1.3-1.5 kW while they run and the clock gives way (2.29-2.36 GHz sustained at
    1.1-1.2 kW board power, box by box) [measured, r5: profiles/r5_side_sync.txt, r5_energy_probe.txt]."""

    def __init__(self, dev):
        import glob
        self.power = self.freq = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hw = f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/"
            for name in ("power1_average", "power1_input"):
                for f in sorted(glob.glob(hw + name)):
                    if self.power is None and open(f).read().split()[0].isdigit():
                        self.power = f
            for f in sorted(glob.glob(hw + "freq1_input")):
                if self.freq is None and open(f).read().split()[0].isdigit():
                    self.freq = f
        except Exception:
            pass

    def __enter__(self):
        import threading
        self.acc, self.stop = [], threading.Event()

        def run():
            while not self.stop.is_set():
                try:
                    self.acc.append((int(open(self.freq).read()) / 1e6 if self.freq else 0.0, int(open(self.power).read()) / 1e6 if self.power else 0.0))
                except Exception:
                    pass
                time.sleep(0.004)
        self.th = threading.Thread(target=run, daemon=True)
        if self.freq or self.power:
            self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.th.is_alive():
            self.th.join()

    def summary(self):
        if len(self.acc) < 4:
            return None
        acc = self.acc[len(self.acc) // 4:]          # (the readings lag the load by a few samples)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        out = {"samples": len(acc)}
        if self.freq:
            out["sclk_mhz_median"] = round(med([a[0] for a in acc]))
        if self.power:
            out["board_power_w_median"] = round(med([a[1] for a in acc]))
        return out


def secondary(solver, dev, n, B, precision, steps, warmup, loc=None, seed=5, readme_first=False, label=None, lanes=1):
    """A short measured run of another configuration (rank 0, N = 1 only): it/s plus the dominant-kernel time."""
    solver.set_unet_precision(precision)
    eng, _, (wf, res, st, k_sq, src) = make_problem(solver, n, B, [n - 62, n // 2] if loc is None else loc, seed, dev, readme_first)
    rmse = torch.zeros(max(steps, warmup), B, device=dev)
    eng.set_option("lanes", lanes)
    try:
        eng.step(wf, res, st, k_sq, src, warmup, rmse_hist=rmse[:warmup])
        torch.cuda.synchronize()
        # two timed regions of `steps` iterations, the faster one reported (both listed): [measured, r5, tools/outlier_probe.py] about one short run in
        # eighty on this pool is 30-40 % slow whatever the kernels (a stall of tens of milliseconds somewhere in the region), and these side
        # measurements are 40-300 steps long.  The HEADLINE region above is timed exactly once, as the contract says
        dts = []
        with Hwmon(dev) as hw:
            for _ in range(2):
                t0 = time.perf_counter()
                eng.step(wf, res, st, k_sq, src, steps, rmse_hist=rmse[:steps])
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
        dt = min(dts)
    finally:
        eng.set_option("lanes", 1)
    flops = 2.0 * unet_macs(n) * B
    return {"workload": label or f"{n}x{n} ring-phantom SoS maps, batch={B}, point source, UNet precision {precision}" + (f", {lanes} pipeline lanes" if lanes > 1 else ""),
            "dtype": DTYPE[precision], "value": round(steps / dt, 2), "unit": "iterations/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 4), "regions_it_per_s": [round(steps / d, 2) for d in dts], "sample_iterations_per_s": round(B * steps / dt, 1),
            "unet_tflops_fp32_equivalent": round(flops * steps / dt / 1e12, 2),
            "residual_rmse_max": float(rmse[steps - 1].max().item()), "hwmon": hw.summary()}


def secondary_dropin_forward(solver, dev, n=256, B=32, K=300, loc=None, seed=0):
    """The call a user of the reference makes: ``IterativeSolver.forward(sos_maps, num_iterations=K)`` with its DEFAULTS (hybridnet.py:654-697: every
    residual kept -- K x B x 2 x N x N floats), timed end to end: get_initials, clear_states, the first residual, K iterations, the histories.  The
    headline times the same K iterations through Engine.step with RMSE rows only; since ABI v7 the residual history costs no copies, so the two
    should agree."""
    from helmnet_amd.phantoms import readme_sos, ring_sos_batch
    solver.set_unet_precision("fp32")
    solver.set_domain_size(n, source_location=[30, n // 2] if loc is None else loc)
    sos_np = ring_sos_batch(n, B, seed=seed)
    if n == 256:
        sos_np[0] = readme_sos()[0]
    sos = torch.from_numpy(sos_np).to(dev)
    out = solver.forward(sos, num_iterations=K)            # warm-up: allocator, tables
    del out
    torch.cuda.synchronize()
    dts = []
    for _ in range(2):
        t0 = time.perf_counter()
        out = solver.forward(sos, num_iterations=K)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
        rm = float(out["residual_norms"][-1].max().item())
        kept = len(out["residuals"])
        del out
    torch.cuda.empty_cache()
    dt = min(dts)
    return {"workload": f"drop-in call: IterativeSolver.forward(sos_maps[{B}x1x{n}x{n}], num_iterations={K}) with default arguments (all {K} residual tensors kept), "
                        "wall clock of the whole call incl. get_initials and the first residual",
            "dtype": "f32", "value": round(K / dt, 2), "unit": "iterations/s", "steps": K, "ms_per_step": round(dt / K * 1e3, 4),
            "regions_it_per_s": [round(K / d, 2) for d in dts], "residuals_kept": kept, "history_gb": round(K * B * 2 * n * n * 4 / 1e9, 2),
            "residual_rmse_max": rm}


def secondary_time_to_tolerance(solver, dev, tol=2e-4, check_every=50, max_iterations=3000):
    """BASELINE configs[4]: the 512 x 512 transcranial phantom at the full 1.87 x contrast with the arc source map (the cfg5 case of tests/test_long_run.py,
    whose iteration counts are pinned against the reference's own 3000-iteration trace), run until the residual RMSE is below `tol`: iterations and
    wall-clock milliseconds to tolerance, fp32 and mixed fp16 UNet / fp32 spectral residual, one map."""
    from helmnet_amd.phantoms import arc_source_map, skull_sos
    sos = torch.from_numpy(skull_sos(512, 1, seed=0)).to(dev)
    src_map = torch.from_numpy(arc_source_map(512)).to(dev)
    runs = []
    for mode in ("fp32", "fp16"):
        solver.set_unet_precision(mode)
        solver.set_domain_size(512, source_map=src_map)
        solver.solve_to_tolerance(sos, tol=tol, max_iterations=check_every, check_every=check_every)   # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = solver.solve_to_tolerance(sos, tol=tol, max_iterations=max_iterations, check_every=check_every)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        runs.append({"unet_precision": mode, "iterations": int(o["iterations"]), "converged": bool(o["converged"]), "ms_to_tolerance": round(dt * 1e3, 2),
                     "ms_per_iteration": round(dt / max(1, o["iterations"]) * 1e3, 4), "final_rmse": float(o["residual_norms"][-1].max().item())})
    solver.set_unet_precision("fp32")
    return {"workload": f"BASELINE configs[4]: 512x512 transcranial phantom (1.87x contrast), arc source map, batch=1, iterate until residual RMSE < {tol} "
                        f"(checked every {check_every} iterations, one 4-byte read-back each)", "unit": "ms to tolerance", "higher_is_better": False,
            "value": runs[0]["ms_to_tolerance"], "dtype": "f32 (value); f16 UNet / f32 spectral residual beside it", "runs": runs}


def secondary_train_step(solver, dev, n=96, B=32, unroll=10, steps=8, warmup=3):
    """one training step (hn_train_grad + hn_adam_step) at the reference's training shape -- 96^2, batch 32, 10
    unrolled iterations (hybridnet.py:385-413) -- from a realistic replay-buffer sample (5 solver iterations in)."""
    from helmnet_amd.engine import pack_weights
    solver.set_unet_precision("fp32")
    eng, _, (wf, res, st, k_sq, src) = make_problem(solver, n, B, [n - 14, n // 2], 5, dev, False)
    eng.step(wf, res, st, k_sq, src, 5)
    src_b = src.repeat(B, 1, 1, 1).contiguous() if src.shape[0] == 1 else src
    w = torch.from_numpy(pack_weights(dict(solver.f.state_dict()))).to(dev)
    m, v, g = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)

    def step(i):
        o = eng.train_grad(w, wf, res, st, k_sq, src_b, unroll, 1e4, grad=g)
        eng.adam_step(w, g, m, v, i + 1, 1e-5, (0.9, 0.95), 1e-8, 1e-6, 1.0)
        return o

    for i in range(warmup):
        o = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        o = step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fwd_flop = 2.0 * unet_macs(n) * B * unroll
    return {"workload": f"training step: {n}x{n}, batch={B}, {unroll} unrolled iterations, forward + backward + Adam (SURVEY 8 f4)",
            "dtype": "f32", "value": round(dt * 1e3, 3), "unit": "ms per training step", "higher_is_better": False, "steps": steps,
            "warmup": warmup, "sample_iterations_per_s": round(B * unroll / dt, 1),
            "approx_tflops_forward_plus_backward": round(3 * fwd_flop / dt / 1e12, 2), "loss": float(o["loss"][0])}


def train_main(args, rank, world, dev, dist):
    """``bench.py --train``: the data-parallel TRAINING step (hybridnet.py:385-413 under Lightning DDP, train.py:103-112) --
    per rank a replay-buffer batch of its own (32 maps at 96^2, 5 solver iterations in), hn_train_grad (10 unrolled iterations, forward +
    backward), ONE all-reduce of the flat 193 KB gradient over RCCL, hn_adam_step.  Weak scaling: 32 maps per GPU.  The replicas start from
    rank 0's weights (broadcast) and stay bit-identical."""
    from helmnet_amd import IterativeSolver
    from helmnet_amd.engine import pack_weights
    from helmnet_amd.training import allreduce_gradients, broadcast_from_rank0
    n, B, K, W, unroll = (96 if args.size == 256 else args.size), args.batch, args.steps, args.warmup, 10
    solver = IterativeSolver.from_exported_weights()
    solver.to(dev)
    eng, _, (wf, res, st, k_sq, src) = make_problem(solver, n, B, [n - 14, n // 2], 100 + rank, dev, False)
    for kv in args.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    eng.step(wf, res, st, k_sq, src, 5)
    src_b = src.repeat(B, 1, 1, 1).contiguous() if src.shape[0] == 1 else src
    w = torch.from_numpy(pack_weights(dict(solver.f.state_dict()))).to(dev)
    m, v, g = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
    broadcast_from_rank0(w, m, v)

    def step(i):
        o = eng.train_grad(w, wf, res, st, k_sq, src_b, unroll, 1e4, grad=g)
        allreduce_gradients(g)
        eng.adam_step(w, g, m, v, i + 1, 1e-5, (0.9, 0.95), 1e-8, 1e-6, 1.0)
        return o

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(W, 2)):
        o = step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        o = step(W + i)
    barrier()
    dt = time.perf_counter() - t0
    same = True
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        w0 = w.clone()
        dist.broadcast(w0, src=0)
        flag = torch.tensor([1.0 if torch.equal(w0, w) else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item() == 1.0)
    if rank == 0:
        fwd_flop = 2.0 * unet_macs(n) * B * unroll
        print(json.dumps({
            "metric": f"training sample-iterations/sec (whole node), {n}^2 domain, batch={B} per GPU x {unroll} unrolled iterations",
            "value": round(world * B * unroll * K / dt, 1), "unit": "sample-iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"training step (SURVEY 8 f4): {n}x{n} ring phantoms, batch={B} per GPU, {unroll} unrolled iterations, forward + "
                                   "backward + gradient all-reduce (one 193 KB bucket, RCCL) + Adam", "options": args.opt,
                       "parallelism": f"dp{world} (each rank its own replay batch; one gradient all-reduce per step)"},
            "approx_tflops_forward_plus_backward": round(world * 3 * fwd_flop * K / dt / 1e12, 2), "loss_rank0": float(o["loss"][0]),
            "replicas_bit_identical_after_run": same}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pin_rank_to_cpus(local: int, nlocal: int):
    """One contiguous slice of the CPUs this process may run on per local rank (the launch threads of N ranks do not migrate over each other)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        if nlocal > 1 and len(cpus) >= nlocal:
            per = len(cpus) // nlocal
            os.sched_setaffinity(0, set(cpus[local * per:(local + 1) * per]))
    except (AttributeError, OSError):
        pass


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N child ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set
    them) BEFORE this process makes any GPU call, relay their output (rank 0 prints the JSON line) and exit with the worst return code.  Children are
    fresh interpreters: nothing that has initialised a GPU is ever re-executed."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n:
        print(f"bench.py: --gpus {n} needs {n} devices on this node, {ndev} visible", file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, alive, stopped = 0, list(procs), []
    while alive:                                     # a rank that fails takes the others (blocked in the rendezvous) with it: exact PIDs only
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and p not in stopped:
                rc = max(rc, abs(code))
                for q in alive:
                    q.terminate()
                    stopped.append(q)
    raise SystemExit(rc)


STEP_COMPULSORY_BYTES_256 = 3751936.0   # read 4[(2+2+1) N^2 + 2 sum N_d^2] + write 4[(2+2) N^2 + 2 sum N_d^2] at N = 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="SoS maps per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--lanes", type=int, default=1, help="hn_step pipeline lanes (sub-batches on parallel streams)")
    ap.add_argument("--precision", "--unet-impl", dest="precision", default="fp32", choices=sorted(PEAK_TFLOPS),
                    help="UNet arithmetic of the HEADLINE run; the reported metric is fp32 (the reference's arithmetic)")
    ap.add_argument("--graph", type=int, default=0, metavar="N",
                    help="replay captured iterations as HIP graphs (1: one iteration per graph, even N: N per graph); default 0 = launch "
                         "every kernel, which measures ~4 %% faster (tools/graph_ab.py)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning knob for A/B runs (hn_set_option): deep=0, side_stream=0, ...")
    ap.add_argument("--settle", type=float, default=None, metavar="SECONDS",
                    help="keep the loop running untimed for this long right before the timed region (clock settling; default 0.3 s "
                         "when --steps < 200, else 0); the iterations it adds are reported as warmup_extra / effective_warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short 512^2 / bf16x3 side measurements")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel time table (stderr)")
    ap.add_argument("--train", action="store_true",
                    help="time the data-parallel TRAINING step instead (96^2, batch 32 per GPU, 10 unrolled iterations, gradient all-reduce over RCCL); "
                         "a separate JSON line, not the headline metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)      # plain `python bench.py --gpus N`: this process never touches a GPU, its N children do
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()       # (counting devices does not initialise the GPU)
    if local >= ndev:
        raise SystemExit(f"bench.py rank {rank}: --gpus {args.gpus} needs {args.gpus} devices on this node, {ndev} visible (LOCAL_RANK={local})")
    pin_rank_to_cpus(local, max(world, 1))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torchrun: same code path for every N
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.train:
        return train_main(args, rank, world, dev, dist)

    from helmnet_amd import IterativeSolver
    from helmnet_amd.distributed import allreduce_residual_norms

    n, B, K, W = args.size, args.batch, args.steps, args.warmup
    prec = args.precision
    loc = [30, n // 2]
    solver = IterativeSolver.from_exported_weights()
    solver.freeze()
    solver.to(dev)
    solver.set_unet_precision(prec)
    eng, sos_np, (wf, res, st, k_sq, src) = make_problem(solver, n, B, loc, rank, dev, rank == 0)   # each rank: its own shard of maps
    eng.set_option("lanes", args.lanes)
    eng.set_option("graph", args.graph)
    for kv in args.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    rmse = torch.zeros(max(K, W, 32), B, device=dev)

    # warm-up, W untimed steps (at least 11): all but the last 8 with every kernel bracketed by events to find the
    # dominant one (>= 3 passes: the first launch of a kernel is not representative), the last 8 as the timed region
    # will run them (graph replays)
    W1 = max(W - 8, 3)
    eng.profile_enable(None)
    eng.step(wf, res, st, k_sq, src, W1, rmse_hist=rmse[:W1])
    torch.cuda.synchronize()
    prof = eng.profile_collect()
    pmin = eng.profile_min()    # shortest launch per kernel: the host cannot keep up with ~70 API calls per step
    # candidates for the roofline block: the two longest kernels of the main chain in the fully bracketed warm-up pass; BOTH are sampled in
    # the timed region and the block describes whichever is longer there (VERDICT r4 #5a; the other one is reported beside it when it is
    # within 5 %).  The side-stream kernels (conv_state*) and the pair bracket are not candidates.
    chain = {k: v for k, v in pmin.items() if not k.startswith("conv_state") and k != "spectral_pair"}
    cand = sorted(chain, key=chain.get, reverse=True)[:2] if chain else ["decode0"]
    cand_ids = [[i for i in range(eng.KERNEL_IDS) if eng.kernel_name(i) == k][0] for k in cand]
    eng.profile_enable([])
    # everything the timed region touches runs once before it: the captured iteration (graph instantiation), the
    # first reduction / collective of the process (code-object loads cost milliseconds on first use)
    eng.step(wf, res, st, k_sq, src, 8, rmse_hist=rmse[:8])   # (the first replays of a freshly instantiated graph are slow)
    # a short run (--steps 20 is 12 ms) would otherwise be timed while the GPU's clocks are still settling (the same kernels
    # measure ~10 % slower in the first tens of milliseconds of a fresh process): keep the loop running, untimed, until
    # 0.3 s of it have passed -- reported as warmup_extra
    allreduce_residual_norms(rmse[0], op="max")
    cap = 8 if K >= 64 else 2                         # at most 8 (short runs: 2) bracketed launches in the timed region: a bracket costs
    stride = max(1, -(-K // cap))                     # ~6 us of stream gap and runs its iteration kernel by kernel instead of as a graph replay

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the settling loop comes LAST, straight in front of the barrier: a few hundred microseconds of idle GPU (the reduction above
    # ends in a host read-back) are enough for the clocks to drop, and the first ~10 ms afterwards then run ~9 % slow --
    # invisible at --steps 300, a tenth of a --steps 20 measurement
    extra, t_w = 0, time.perf_counter()
    settle_s = args.settle if args.settle is not None else (0.3 if K < 200 else 0.0)
    while time.perf_counter() - t_w < settle_s:
        eng.step(wf, res, st, k_sq, src, 32, rmse_hist=rmse[:min(32, rmse.shape[0])] if rmse.shape[0] >= 32 else None)
        torch.cuda.synchronize()
        extra += 32
    pair_id = [i for i in range(eng.KERNEL_IDS) if eng.kernel_name(i) == "spectral_pair"][0]
    eng.profile_enable(cand_ids + [pair_id])          # host-side state only: the two candidate kernels and the spectral pair, sampled
    eng.profile_stride(stride)
    replays0, eager0 = eng.counter("graph_replays"), eng.counter("eager_iterations")

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
    collected = eng.profile_collect()
    pair_ms, pair_cnt = collected.get("spectral_pair", (0.0, 0))
    replays, eager = eng.counter("graph_replays") - replays0, eng.counter("eager_iterations") - eager0
    eng.profile_enable([])
    eng.profile_stride(1)

    if rank == 0:
        macs, byts_k = kernel_macs(n), kernel_bytes(n)
        total_flops = 2.0 * unet_macs(n) * B
        per_step = max(1, args.lanes if B >= 2 * args.lanes else 1)   # hn_step may split the batch over pipeline lanes
        dc_valu = int(dict(kv.split("=", 1) for kv in args.opt).get("dc_valu", "4")) if prec == "fp32" and n >= 256 else 0
        peak = PEAK_TFLOPS[prec]

        def kernel_roof(name, ms, cnt):
            """The roofline block of one kernel from its sampled launches in the timed region."""
            avg_s = ms / cnt * 1e-3
            traffic, traffic_src = measured_traffic(name, n, B, prec)
            if name not in macs:   # a spectral pass: HBM-bound
                byts = spectral_bytes(n) * B / per_step
                ach = byts / avg_s / 1e9
                return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt, "bytes_per_launch": byts}
            flops = 2.0 * macs[name] * B / per_step
            hbm = byts_k[name] * B / per_step
            ach_f, ach_b = flops / avg_s / 1e12, hbm / avg_s / 1e9
            r = {"kernel": name, "bound": "mfma", "achieved": round(ach_f, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach_f / peak, 4),
                 "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
                 "samples_per_launch": B // per_step, "flops_per_launch": flops, "flops_credited_per_launch": flops, "product_terms_per_flop": TERMS[prec],
                 "hbm_view": {"bytes_per_launch": hbm, "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 4)}}
            on_vector_pipe = (name in ("decode0", "inc") and dc_valu >= 1) or (name == "conv_signal0" and dc_valu in (2, 4)) or name == "inc_conv_signal0"
            if on_vector_pipe:
                # hn_dca.hip / hn_dcv.hip: the level-0 DoubleConvs run on the packed fp32 VECTOR FMA, whose peak on gfx950 equals the fp32
                # matrix peak (157.3 TFLOP/s, 64 FLOP / clk / SIMD); "bound" keeps the schema's compute label
                r["pipe"] = "v_pk_fma_f32 (fp32 vector FMA; peak = fp32 MFMA peak)"
                ex = kernel_macs_executed(n, name)
                if ex is not None:   # credited = the reference layers' FLOPs; executed = what the kernel's lanes compute (mid-tile halo, composed final layer)
                    fe = 2.0 * ex * B / per_step
                    r["flops_executed_per_launch"] = fe
                    r["executed"] = {"achieved": round(fe / avg_s / 1e12, 2), "unit": "TFLOP/s", "frac": round(fe / avg_s / 1e12 / peak, 4)}
            if ach_b / HBM_PEAK_GBS > ach_f * TERMS[prec] / peak:   # 16-bit modes: the level-0 DoubleConvs become HBM-bound
                r.update({"bound": "hbm", "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 4),
                          "mfma_view": {"achieved": round(ach_f, 2), "peak": peak, "unit": "TFLOP/s (fp32-equivalent)"}})
                r.pop("hbm_view")
            return r

        sampled = sorted(((k,) + collected[k] for k in cand if collected.get(k, (0.0, 0))[1]), key=lambda t: -(t[1] / t[2]))
        roof = None
        if sampled:
            roof = kernel_roof(*sampled[0])
            roof["selection"] = ("the longer (timed-region average) of the two longest main-chain launches of the fully bracketed warm-up pass: " + ", ".join(cand))
            if roof["kernel"] == "inc_conv_signal0":
                roof["merged_launch_note"] = ("this launch is TWO reference layers -- inc and conv_signal_0 as one grid whose second half waits, tile by tile, on flags the first "
                                              "half publishes (HN_OPT_DC_PAIR, DESIGN.md 4.1a); launched separately (--opt dc_pair=0) they take 52 + 66 us (0.51 / 0.53 of peak, "
                                              "profiles/r5_kernel_stats.csv history in DESIGN.md 4); the longest SINGLE-layer kernel is the runner_up, decode0")
            if len(sampled) > 1:
                other = kernel_roof(*sampled[1])
                roof["runner_up"] = other                       # always reported; "tie" says whether it is within 5 % of the longest
                roof["tie"] = other["avg_launch_us"] >= 0.95 * roof["avg_launch_us"]
            # the whole step against the same peak: total credited FLOPs of one iteration (every layer of the UNet) over the measured time per step
            roof["step"] = {"flops_per_step": total_flops, "ms_per_step": round(dt / K * 1e3, 4), "achieved": round(total_flops * K / dt / 1e12, 2), "peak": peak,
                            "unit": "TFLOP/s", "frac": round(total_flops * K / dt / 1e12 / peak, 4),
                            "note": "whole iteration: UNet FLOPs of the reference layers / time per step (the spectral pair's 0.5 GFLOP not counted)"}
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (torchrun pins OMP threads to 1 per rank)
            cpu = cpu_baseline(torch.from_numpy(sos_np), n, loc)
        final_rmse = rmse[K - 1].float().cpu().numpy()
        cfg_name = {(256, 32): "BASELINE configs[1]", (512, 16): "BASELINE configs[3]"}.get((n, B), "non-BASELINE shape")
        line = {
            "metric": f"solver iterations/sec (whole node), {n}^2 domain batch={B}",
            "value": round(world * K / dt, 2),
            "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W, "warmup_extra": W1 + 8 + extra - W, "effective_warmup": W1 + 8 + extra,
            "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[prec], "data": "synthetic",
            "config": {"workload": f"{n}x{n} ring-phantom SoS maps, batch={B} per GPU, point source {loc}, "
                                   f"shipped jcp checkpoint weights, UNet precision {prec} ({cfg_name})",
                       "batch_per_gpu": B, "domain": n, "lanes": args.lanes, "unet_precision": prec, "options": args.opt,
                       "iteration_launch": {"graph_replays": replays, "kernel_by_kernel": eager},
                       "parallelism": f"dp{world} (batch shards, no data-path collective)"},
            "sample_iterations_per_s": round(world * B * K / dt, 1),
            "unet_tflops": round(total_flops * K / dt / 1e12, 2),
            "residual_rmse_after_timed_steps": {"median": float(np.median(final_rmse)), "max": float(worst.max().item()),
                                                "iterations_from_zero_wavefield": W1 + 8 + extra + K},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        # whole-step HBM view: compulsory bytes of one fused sample-iteration x batch x iterations/s against the HBM peak.
        # The step is compute-bound (fused arithmetic intensity 281 FLOP/B): at the fp32 peak the figure could reach ~7 %.
        step_bytes = STEP_COMPULSORY_BYTES_256 * (n / 256.0) ** 2 * B
        step_gbs = step_bytes * world * K / dt / 1e9
        line["step_hbm"] = {"compulsory_bytes_per_step": step_bytes, "achieved": round(step_gbs / world, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(step_gbs / world / HBM_PEAK_GBS, 4),
                            "note": "per GPU; whole iteration fused would move 3,751,936 B per sample at 256^2 (SURVEY 8d); compute-limited ceiling ~0.07"}
        if roof is not None and prec == "fp32":
            roof["flops_note"] = ("achieved / frac use the CREDITED FLOPs: the reference layers' (decode0: conv 16->8, conv 8->8, 1x1 8->2 = 228.6 MFLOP per sample "
                                  "at 256^2); `executed` is what the kernel's lanes compute (conv1 on the 18 x 66 mid tile of every 16 x 64 output tile; decode0's "
                                  "final 3x3 + 1x1 composed into one 2-channel 3x3)")
        # every level-0 kernel of the main chain: shortest event-bracketed launch of the warm-up pass.  Durations only (VERDICT r4 #5c): each carries
        # ~5-8 us of event overhead, so no rate is derived from them -- the rocprofv3 durations of the same command are in profiles/r5_kernel_stats.csv
        line["level0_kernels_note"] = ("shortest launch per kernel in the warm-up pass in which EVERY kernel is bracketed by an event pair; each figure carries ~5-8 us of "
                                       "event overhead (compare roofline.avg_launch_us, sampled in the timed region); rocprofv3 durations: profiles/r5_kernel_stats.csv")
        line["level0_kernels"] = [{"kernel": k, "bracketed_us": round(pmin[k] * 1e3, 2), "gflop_credited": round(2.0 * macs[k] * B / per_step / 1e9, 3)}
                                  for k in ("inc", "conv_signal0", "inc_conv_signal0", "down0", "up0", "decode0") if k in pmin and k in macs and pmin[k] > 0]
        # secondary line for the HBM-bound part of the path (north_star: "achieved HBM GB/s for the FFT path"):
        # compulsory bytes of get_residual (5 planes per sample) over the shortest bracketed launches of the two
        # spectral kernels in the fully bracketed warm-up pass
        sp_us = (pmin.get("spectral_cols", 0.0) + pmin.get("spectral_rows", 0.0)) * 1e3
        if pair_cnt:   # both passes under one event pair, sampled in the timed region like the dominant kernel
            sp_us = pair_ms / pair_cnt * 1e3
        if sp_us > 0:
            sp_bytes = spectral_bytes(n) * B
            line["hbm_path"] = {"kernels": ["spectral_cols", "spectral_rows"], "bound": "hbm", "bytes_per_step": sp_bytes,
                                "timing": f"one event pair around both kernels, {pair_cnt} samples in the timed region" if pair_cnt else "shortest bracketed launches of the warm-up pass",
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
        if world == 1 and not args.no_secondary and (n, B, prec) == (256, 32, "fp32"):
            # driver-visible side measurements: same process, a few seconds each
            line["secondary"] = [secondary(solver, dev, 256, 32, "fp32", 300, 20, loc=loc, seed=rank, readme_first=True,
                                           label="the HEADLINE workload again, 300 timed steps after 20 (VERDICT r3 #10: the driver's --steps 20 "
                                                 "region is 10 ms; this is the same loop over 150 ms in the same process)"),
                                 secondary(solver, dev, 512, 16, "fp32", 40, 10),
                                 secondary(solver, dev, 256, 32, "bf16x3", 60, 10),
                                 secondary(solver, dev, 512, 16, "fp16", 40, 10),
                                 secondary(solver, dev, 256, 64, "fp32", 60, 10),   # throughput beyond the headline batch (one chain: since r5 faster than two lanes at 64, tools/ab_inproc.py lanes 1,2 --batch 64)
                                 secondary_train_step(solver, dev),
                                 secondary_dropin_forward(solver, dev, 256, 32, 300, loc=loc, seed=rank),
                                 secondary_time_to_tolerance(solver, dev)]
            hw0 = line["secondary"][0].get("hwmon")
            if hw0 and hw0.get("sclk_mhz_median"):
                f = hw0["sclk_mhz_median"] / 2400.0
                line["clock"] = {**hw0, "nominal_mhz": 2400, "frac_of_nominal": round(f, 4), "fp32_peak_at_sustained_clock_tflops": round(157.3 * f, 1),
                                 "region": "the 300-step repeat of the headline loop (secondary[0]); the --steps 20 region is too short to sample",
                                 "note": "roofline.peak is the NOMINAL 157.3 TFLOP/s (2.4 GHz x 256 CUs x 256 FLOP/clk); the loop sustains this clock instead: its "
                                         "level-0 kernels draw 1.3-1.5 kW while they run and the power management lowers the clock (DESIGN.md 5)"}
            line["secondary_note"] = ("bf16x3 is an fp32-accurate EMULATION (3-term bf16 split, 6 product terms, fp32 accumulate), "
                                      "fp16 is the mixed-precision configuration of BASELINE configs[4]; neither replaces the fp32 headline")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
