#!/usr/bin/env python3
"""bench.py -- k-mers counted per second at K=31 on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path -- hsk_count_device(): minimizer/supermer parse, [RCCL
supermer exchange], per-task k-mer extraction, LSD radix sort, merge-count + [L,U] filter -- over
one batch of synthetic reads that is ALREADY RESIDENT IN HBM (generated on the device), with the
result left in HBM.  Workload at N=1 is BASELINE.json configs[1]: S-reads(G = 312.5 Mbp, c = 32),
i.e. 10 Gbp of error-free 150-bp reads, 8.0e9 31-mers, L=15 U=40 (reference Makefile defaults).
For N > 1 every rank holds 10 Gbp of reads sampled from ONE genome of N x 312.5 Mbp (configs[2] at
N=8: 80 Gbp), so the minimizer exchange is real; scaling is weak.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (radix scatter pass): algorithmic
bytes per launch (2 x 8 B x keys) / average launch duration measured with HIP events on the launch
stream inside the timed region.  `cpu_baseline` times the CPU path on the host cores on a bounded
sample (1/`--cpu-div` of the workload): the real reference binary built by oracle/build_ref.sh when
it is present and runnable ("reference"), else the C restatement in oracle/ ("port").
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME_PER_GPU = 312_500_000
READ_LEN = 150
COVERAGE = 32
K, M, L, U = 31, 17, 15, 40
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; 1.0 = BASELINE config)")
    ap.add_argument("--ntasks", type=int, default=0)
    ap.add_argument("--cpu-div", type=int, default=200, help="cpu_baseline sample = workload / this")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ext", type=int, default=0)
    ap.add_argument("--k", type=int, default=31, help="k-mer size (informational runs; the headline metric is K=31)")
    return ap.parse_args()


def write_fasta_sample(path, seqs, width=80):
    with open(path, "w") as f, open(path + ".fai", "w") as fai:
        off = 0
        for i, s in enumerate(seqs):
            hdr = ">r%d\n" % i
            f.write(hdr)
            off += len(hdr)
            fai.write("r%d\t%d\t%d\t%d\t%d\n" % (i, len(s), off, width, width + 1))
            body = "\n".join(s[j:j + width] for j in range(0, len(s), width)) + "\n"
            f.write(body)
            off += len(body)


def cpu_baseline(ctx, genome_len, nreads, seed, ncores):
    """Times the CPU path on a bounded sample of the same workload; returns the JSON object."""
    from oracle import hsk_oracle as O
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed)
    packed = ctx.d2h(dp, nb)
    ctx.synth_free(dp, do, dl)
    nbr = (READ_LEN + 3) // 4
    off = np.arange(nreads, dtype=np.uint64) * np.uint64(nbr)
    lens = np.full(nreads, READ_LEN, dtype=np.uint32)
    nk = nreads * (READ_LEN - K + 1)
    sample = "S-reads(G=%d, c=%d): %d x %d-bp reads, %d k-mers" % (genome_len, COVERAGE, nreads, READ_LEN, nk)
    # --- the C restatement (OpenMP over reads and tasks)
    t0 = time.time()
    ores = O.count(packed, off, lens, k=K, m=M, L=L, U=U, ntasks=max(ncores * 2, 8), fast=True)
    t_port = time.time() - t0
    port = {"value": nk / t_port, "unit": "k-mers/s", "cores": ncores, "kind": "port", "sample": sample,
            "seconds": round(t_port, 3), "entries": int(ores.cnt.size)}
    # --- the real reference, if its binary travelled and runs here
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "k31", "hysortk_ref")
    mpiexec = "/opt/conda/bin/mpiexec"
    if os.path.exists(ref_bin) and os.path.exists(mpiexec):
        try:
            lut = np.frombuffer(b"ACGT", dtype=np.uint8)
            pk = packed.reshape(nreads, nbr)
            codes = np.stack([(pk >> 6) & 3, (pk >> 4) & 3, (pk >> 2) & 3, pk & 3], axis=2).reshape(nreads, nbr * 4)[:, :READ_LEN]
            seqs = [r.tobytes().decode() for r in lut[codes]]
            tmp = tempfile.mkdtemp(prefix="hsk_cpu_")
            fa = os.path.join(tmp, "sample.fa")
            write_fasta_sample(fa, seqs)
            env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")
            best = None
            # the reference scales with ranks ~ NUMA domains (README.md:46): try ranks x threads layouts
            layouts = [(r, max(1, ncores // r)) for r in (max(1, ncores // 2), max(1, ncores // 4), max(1, ncores // 8)) if r >= 1]
            for ranks, thr in dict.fromkeys(layouts):
                if ranks > nreads:
                    continue
                e = dict(env, OMP_NUM_THREADS=str(thr))
                p = subprocess.run([mpiexec, "-n", str(ranks), ref_bin, fa], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   text=True, timeout=120)
                m = re.search(r"Overall kmer counting \(Excluding I/O\):\s*\n\s*total time \(user seconds\):\s*([0-9.]+)", p.stdout)
                if p.returncode == 0 and m:
                    t = float(m.group(1))
                    if best is None or t < best[0]:
                        best = (t, ranks, thr)
            if best:
                ref = {"value": nk / best[0], "unit": "k-mers/s", "cores": ncores, "kind": "reference",
                       "sample": sample + "; reference built with L=1 U=65535 (filter off), RADULS, %d ranks x %d threads, its own "
                       "'Overall kmer counting (Excluding I/O)' timer" % (best[1], best[2]),
                       "seconds": best[0], "port": port}
                return ref
        except Exception as e:  # the baseline must never break the bench line
            port["reference_error"] = str(e)[:200]
    return port


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: this process (which has imported neither torch nor anything that
    touches HIP) starts the N ranks as a FRESH child -- python -m torch.distributed.run, one process per GPU -- relays
    rank 0's JSON line and returns the child's exit code."""
    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except Exception:
        ndev = 0
    if ndev < a.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d MI355X, this machine shows %d HIP device(s); there is no CPU fallback\n" % (a.gpus, a.gpus, ndev))
        return 2
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line:
        print(line)
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 3
    return rc


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HSK_FORCE_DEVICE") is not None:          # debugging aid: several ranks on one GPU
        local = int(os.environ["HSK_FORCE_DEVICE"])
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))                                # one command runs all ranks (reference: mpiexec -n N ./hysortk, README.md:39)
    if world != a.gpus:
        a.gpus = world
    import torch
    import hysortk_amd as H
    from hysortk_amd import dist as hdist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    comm = None
    if world > 1:
        # host-side collectives (id broadcast, barriers, max-over-ranks) go over gloo; the supermer
        # payload moves inside libhsk.so with RCCL send/recv over xGMI
        comm = hdist.Comm(backend="gloo")

    genome_len = int(GENOME_PER_GPU * a.scale) * world
    nreads = int(GENOME_PER_GPU * a.scale) * COVERAGE // READ_LEN
    KK = a.k
    nk_rank = nreads * (READ_LEN - KK + 1)
    seed = 20251003

    ctx = H.Context(K=KK, M=M, L=L, U=U, EXT=a.ext, ntasks=a.ntasks, device=local, profile=True, keep_device=True)
    ctx.comm_init(comm)
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed, first_read=rank * nreads)

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    info = None
    for _ in range(a.warmup):
        info = ctx.count_device(dp, nb, do, dl, nreads, rid_base=rank * nreads).info
    ctx.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    phase = {}
    for _ in range(a.steps):
        info = ctx.count_device(dp, nb, do, dl, nreads, rid_base=rank * nreads).info
        for k_, v in info.items():
            if k_.startswith("ms_"):
                phase[k_] = phase.get(k_, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats(reset=True)
    if comm is not None:
        dt = comm.allreduce_max(dt)
    if rank == 0:
        total_kmers = nk_rank * world * a.steps
        value = total_kmers / dt
        launches = max(int(st["scatter_launches"]), 1)
        avg_ms = st["scatter_ms"] / launches
        bytes_per_launch = st["scatter_bytes"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes per launch (rocprofv3), if collected
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("onesweep_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "k-mers counted/sec at K=%d" % KK, "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("u64" if KK <= 32 else "u128") + ("" if not a.ext else "+u64 payload"), "data": "synthetic",
            "config": {"workload": "S-reads(G=%d bp x %d GPU, c=%d): %d x %d-bp reads per GPU = %.3g bp, %d k-mers per GPU" % (
                int(GENOME_PER_GPU * a.scale), world, COVERAGE, nreads, READ_LEN, nreads * READ_LEN, nk_rank),
                "K": KK, "M": M, "L": L, "U": U, "EXT": a.ext, "ntasks": info["ntasks"], "scale": a.scale,
                "input": "resident in HBM", "output": "left in HBM (entries=%d on rank 0)" % info.get("n", -1),
                "exchange": "RCCL send/recv all-to-all-v" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "onesweep_multi_kernel (radix scatter pass, 8 tasks per launch, one per XCD; with the first pass fused into the expand this is the one remaining pass, over chunk tiles)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches": int(st["scatter_launches"]),
                         "avg_launch_ms": avg_ms, "bytes_per_launch": bytes_per_launch,
                         # kind-1 events: expand_scatter_kernel (expand + first scatter pass; bytes = keys written) when the fused
                         # path runs, the histogram kernel otherwise
                         "expand_scatter_avg_launch_ms": (st["hist_ms"] / max(int(st["hist_launches"]), 1)) if st["hist_ms"] else None,
                         "expand_scatter_written_GBs": (st["hist_bytes"] / max(st["hist_ms"], 1e-9) / 1e6) if st["hist_ms"] else None,
                         "agg_GBs": (st["agg_bytes"] / max(st["agg_ms"], 1e-9) / 1e6) if st["agg_ms"] else None,
                         "agg_avg_launch_ms": (st["agg_ms"] / max(int(st["agg_launches"]), 1)) if st["agg_ms"] else None},
            "path_stats": {k_: int(st[k_]) for k_ in ("fused_tasks", "redone_tasks", "agg_retried_tasks", "parse_fallbacks", "heavy_tasks", "onepass_misses") if k_ in st},
            "phases_ms_per_step": {k_: v / a.steps for k_, v in sorted(phase.items())},
            "whole_path_algorithmic_GBs": ((152.3 if KK <= 32 and not a.ext else (464.4 if KK > 32 else 304.3)) * nk_rank) / (phase.get("ms_total", 0) / a.steps * 1e-3) / 1e9 if phase.get("ms_total") else None,
        }
        if world == 1 and not a.no_cpu:
            ncores = os.cpu_count() or 1
            div = max(a.cpu_div, 1)
            g_s = max(int(GENOME_PER_GPU * a.scale) // div, 10000)
            out["cpu_baseline"] = cpu_baseline(ctx, g_s, g_s * COVERAGE // READ_LEN, seed + 1, ncores)
        print(json.dumps(out))
        sys.stdout.flush()
    ctx.synth_free(dp, do, dl)
    ctx.close()
    if comm is not None:
        comm.barrier()
        comm.destroy()


if __name__ == "__main__":
    main()
