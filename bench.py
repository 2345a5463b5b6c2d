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

`--gpus N` (N > 1) without a launcher starts the N ranks itself (one fresh child:
python -m torch.distributed.run, one process per GPU); under a launcher (WORLD_SIZE set) this file is a rank.

Rank 0 prints ONE JSON line:
  value / ms_per_step   the device-resident rate above (inputs resident in HBM when the timed region starts, as the bench
                        contract asks; `value` never includes PCIe).  The same figure again under `device_resident`.
  host_to_host          (N = 1) the metric exactly as SURVEY 8(d) and the reference's own timer define it (src/hysortk.cpp:58,91,
                        "Overall kmer counting (Excluding I/O)"): wall time of hsk_count() from a DnaBuffer in (pinned) host RAM
                        to the KmerListS entries in host RAM, L=15/U=40, with its PCIe shares
  roofline              the TIME-DOMINANT kernel of this run: algorithmic bytes per launch / average launch duration from HIP
                        events on the launch stream inside the timed region, against the 8 TB/s spec and against the copy rate
                        measured in this run (hsk_copy_peak: hand-written 16-byte-per-lane copy); .whole_path: the step against
                        the PMC-measured traffic of profiles/traffic.json
  kernels               the kernels that make up the step, ms per step and what bounds each
  variants              (N = 1) short legs of the other record shapes and plans, 3 steps each, device-resident: K=51 (two-word
                        keys), EXTENSION=1, the same workload without the LDS aggregation, with the reference's own algorithm
                        (LSD over all key bytes + merge-count: the path BASELINE.json's north_star names, against SURVEY 8(d)'s
                        fixed 152.3 B per k-mer), and uniform random reads (every k-mer once, L=1: the worst case for output)
  cpu_baseline          the CPU path on the host cores on a bounded sample (1/`--cpu-div` of the workload): the real reference
                        binary built by oracle/build_ref.sh with the same L=15/U=40 when it travelled ("reference"; best of
                        several ranks x threads layouts), else the C restatement ("port").  The GPU counts the same sample and
                        its histogram must equal the reference's (`sample_check`).
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
    ap.add_argument("--cpu-div", type=int, default=20, help="cpu_baseline sample = workload / this")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host leg (N = 1)")
    ap.add_argument("--no-variants", action="store_true", help="skip the legs of the other record shapes and plans (N = 1)")
    ap.add_argument("--legs", default="", help="comma-separated: run only these of the informational legs (variant names, multi_rank, large, first_calls); default all")
    ap.add_argument("--error-rate", type=float, default=0.0, help="substitution errors per base in the synthetic reads (informational runs; the headline workload is error-free)")
    ap.add_argument("--ext", type=int, default=0)
    ap.add_argument("--k", type=int, default=31, help="k-mer size (informational runs; the headline metric is K=31)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): test infrastructure timed beside the product, never on its path
# ---------------------------------------------------------------------------------------------------------------------
def write_fasta_sample(path, packed, nreads):
    """One record per line (header, 150 bases), vectorised; .fai beside it."""
    nbr = (READ_LEN + 3) // 4
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    rec = np.empty((nreads, 3 + READ_LEN + 1), dtype=np.uint8)
    rec[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
    rec[:, 3 + READ_LEN] = 10
    step = 1 << 18
    for a in range(0, nreads, step):
        pk = packed[a * nbr:(a + step) * nbr].reshape(-1, nbr)
        codes = np.stack([(pk >> 6) & 3, (pk >> 4) & 3, (pk >> 2) & 3, pk & 3], axis=2).reshape(pk.shape[0], nbr * 4)[:, :READ_LEN]
        rec[a:a + pk.shape[0], 3:3 + READ_LEN] = lut[codes]
    with open(path, "wb") as f:
        f.write(rec.tobytes())
    w = rec.shape[1]
    with open(path + ".fai", "w") as f:
        for a in range(0, nreads, step):
            f.write("".join("r\t%d\t%d\t%d\t%d\n" % (READ_LEN, i * w + 3, READ_LEN, READ_LEN + 1) for i in range(a, min(a + step, nreads))))


def parse_histogram(text):
    m = re.search(r"#count\tnumkmers\n((?:\d+\t\d+\n)*)", text)
    return {int(l.split("\t")[0]): int(l.split("\t")[1]) for l in m.group(1).splitlines()} if m else None


def cpu_baseline(H, local, genome_len, nreads, seed, ncores, fraction):
    """Times the CPU path on a bounded sample of the same workload (same K, M, L, U as the GPU leg) and counts the sample on the
    GPU as well: the three histograms (reference binary, C restatement, HIP) must be equal.  Returns the JSON object."""
    from oracle import hsk_oracle as O
    ctx = H.Context(K=K, M=M, L=L, U=U, device=local)
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed)
    gres = ctx.count_device(dp, nb, do, dl, nreads)
    gpu_hist = {int(c): int(v) for c, v in enumerate(gres.histo) if v and c >= 1}
    gpu_entries = len(gres)
    del gres
    packed = ctx.d2h(dp, nb)
    ctx.synth_free(dp, do, dl)
    ctx.close()
    nbr = (READ_LEN + 3) // 4
    off = np.arange(nreads, dtype=np.uint64) * np.uint64(nbr)
    lens = np.full(nreads, READ_LEN, dtype=np.uint32)
    nk = nreads * (READ_LEN - K + 1)
    sample = "S-reads(G=%d, c=%d): %d x %d-bp reads, %d k-mers = 1/%d of the GPU workload, L=%d U=%d" % (genome_len, COVERAGE, nreads, READ_LEN, nk, round(1 / fraction), L, U)
    # --- the C restatement (OpenMP over reads and tasks)
    t0 = time.time()
    ores = O.count(packed, off, lens, k=K, m=M, L=L, U=U, ntasks=max(ncores * 2, 8), fast=True)
    t_port = time.time() - t0
    port_hist = parse_histogram(O.histogram_text(ores.cnt)) or {}
    n_entries = int(ores.cnt.size)
    del ores
    check = {"gpu_entries": gpu_entries, "port_entries": n_entries, "gpu_equals_port": gpu_hist == port_hist and gpu_entries == n_entries}
    port = {"value": nk / t_port, "unit": "k-mers/s", "cores": ncores, "kind": "port", "sample": sample, "sample_fraction": fraction,
            "seconds": round(t_port, 3), "entries": n_entries, "sample_check": check}
    # --- the real reference, if its binary travelled and runs here
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "k31b", "hysortk_ref")          # oracle/build_ref.sh k31b 31 17 15 40 0 2
    mpiexec = "/opt/conda/bin/mpiexec"
    if os.path.exists(ref_bin) and os.path.exists(mpiexec):
        try:
            tmp = tempfile.mkdtemp(prefix="hsk_cpu_")
            fa = os.path.join(tmp, "sample.fa")
            write_fasta_sample(fa, packed, nreads)
            env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")
            # the reference scales with ranks ~ NUMA domains (README.md:46): ranks x threads = cores
            layouts = [(r, max(1, ncores // r)) for r in (8, 16, 32, 64) if r <= ncores]
            if ncores < 16:
                layouts.append((max(1, ncores // 2), 2))
            tried, best, ref_hist = [], None, None
            t_leg = time.time()

            def run_ref(ranks, thr, bind):
                nonlocal best, ref_hist
                e = dict(env, OMP_NUM_THREADS=str(thr))
                cmd = [mpiexec] + (["-bind-to", bind] if bind else []) + ["-n", str(ranks), ref_bin, fa]
                p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
                m = re.search(r"Overall kmer counting \(Excluding I/O\):\s*\n\s*total time \(user seconds\):\s*([0-9.]+)", p.stdout)
                hist = parse_histogram(p.stdout)
                ok = p.returncode == 0 and m is not None and hist == gpu_hist     # oversubscribed, the reference silently drops entries
                tried.append({"ranks": ranks, "threads": thr, "bind": bind or "none", "seconds": float(m.group(1)) if m else None,
                              "entries": sum(hist.values()) if hist else -1, "histogram_equals_gpu": hist == gpu_hist})
                if ok and (best is None or float(m.group(1)) < best[0]):
                    best = (float(m.group(1)), ranks, thr, bind or "none")
                    ref_hist = hist

            for ranks, thr in dict.fromkeys(layouts):
                if ranks <= nreads and time.time() - t_leg < 100:
                    run_ref(ranks, thr, None)
            if best:                                   # the best layout again with its ranks pinned (hydra + hwloc): NUMA placement is what the layouts differ by
                b_ranks, b_thr = best[1], best[2]
                for bind in ("hwthread:%d" % b_thr, "core:%d" % max(1, b_thr // 2)):
                    if time.time() - t_leg < 140:
                        run_ref(b_ranks, b_thr, bind)
            for f_ in (fa, fa + ".fai"):
                os.remove(f_)
            os.rmdir(tmp)
            if best:
                check["reference_entries"] = sum(ref_hist.values())
                check["gpu_equals_reference"] = ref_hist == gpu_hist
                check["histogram_bins"] = len(gpu_hist)
                return {"value": nk / best[0], "unit": "k-mers/s", "cores": ncores, "kind": "reference", "sample_fraction": fraction,
                        "sample": sample + "; reference built with the same L / U, RADULS, best of the layouts below: %d ranks x %d threads (mpiexec -bind-to %s), "
                        "its own 'Overall kmer counting (Excluding I/O)' timer (src/hysortk.cpp:58,91)" % (best[1], best[2], best[3]),
                        "seconds": best[0], "entries": n_entries, "layouts": tried, "sample_check": check, "port": port}
            port["reference_layouts"] = tried
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
    if ndev < a.gpus and os.environ.get("HSK_FORCE_DEVICE") is None:      # (HSK_FORCE_DEVICE + HSK_RCCL_LIB: the ranks share one GPU over the tests' stand-in transport)
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


def run_variant(H, local, name, note, KK, ext, plan, Lv, Uv, genome_len, nreads, seed, error_rate, steps, ref_bytes_per_kmer):
    """One short device-resident leg of another record shape / plan: k-mers/s, ms per step, the scatter passes' GB/s."""
    ctx = H.Context(K=KK, M=M, L=Lv, U=Uv, EXT=ext, device=local, profile=True, keep_device=True, plan=plan)
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed, error_rate=error_rate)
    # the FIRST call of a fresh context, timed on its own: the plan is chosen inside the call (estimate_plan), so what it costs beyond the
    # steady state is the context's memory pools growing (hipMalloc of tens of GB), not a plan tried and abandoned
    ctx.stats(reset=True)
    t0 = time.perf_counter()
    r = ctx.count_device(dp, nb, do, dl, nreads)
    first_ms = (time.perf_counter() - t0) * 1e3
    first_dev_ms = r.info["ms_total"]
    st1 = ctx.stats(reset=True)
    del r
    r = ctx.count_device(dp, nb, do, dl, nreads)          # second call: the pools have the sizes of this path now
    del r
    ctx.stats(reset=True)
    t0 = time.perf_counter()
    info = None
    for _ in range(steps):
        r = ctx.count_device(dp, nb, do, dl, nreads)
        info = dict(r.info)
        del r
    dt = (time.perf_counter() - t0) / steps
    st = ctx.stats(reset=True)
    ctx.synth_free(dp, do, dl)
    ctx.close()
    nk = nreads * (READ_LEN - KK + 1)
    out = {"name": name, "what": note, "K": KK, "EXT": ext, "L": Lv, "U": Uv, "plan": plan or "default", "error_rate": error_rate, "kmers": nk, "steps": steps,
           "value": nk / dt, "unit": "k-mers/s", "ms_per_step": dt * 1e3, "device_ms_total": info["ms_total"], "entries": info.get("n"), "ntasks": info["ntasks"],
           "first_call_ms": first_ms, "first_call_device_ms": first_dev_ms, "first_call_combine_launches": int(st1.get("combine_launches", 0)),
           "first_call_note": "fresh context, no warm-up call: host wall clock incl. the memory pools growing; first_call_device_ms = the device time of that call (HIP events)",
           "combine_launches_per_step": st.get("combine_launches", 0) / steps,
           "phases_ms": {k_: round(v, 2) for k_, v in info.items() if k_.startswith("ms_")},
           "scatter_pass": {"launches_per_step": st["scatter_launches"] / steps, "ms_per_step": st["scatter_ms"] / steps,
                            "GBs": (st["scatter_bytes"] / (st["scatter_ms"] * 1e-3) / 1e9) if st["scatter_ms"] > 0 else None},
           "path_stats": {k_: int(st[k_]) for k_ in ("fused_tasks", "redone_tasks", "agg_retried_tasks") if k_ in st}}
    if ref_bytes_per_kmer:
        out["reference_algorithm_GBs"] = ref_bytes_per_kmer * nk / dt / 1e9
        out["reference_algorithm_frac_of_hbm_peak"] = out["reference_algorithm_GBs"] / HBM_PEAK_GBS
        out["reference_algorithm_bytes_per_kmer"] = ref_bytes_per_kmer
    return out


def large_input_leg(H, local, gbp, seed):
    """The whole input of BASELINE configs[2] (80 Gbp: what the 8-GPU configuration counts) in ONE call on ONE GPU, same L / U as the headline, the
    list left in HBM: what 288 GB of HBM hold.  One untimed call (the pools grow), two timed ones."""
    G = int(gbp * 1e9) // COVERAGE
    NR = G * COVERAGE // READ_LEN
    ctx = H.Context(K=K, M=M, L=L, U=U, device=local, profile=True, keep_device=True)
    dp, nb, do, dl = ctx.synth_reads(G, READ_LEN, NR, seed)
    r = ctx.count_device(dp, nb, do, dl, NR)
    del r
    ts, info = [], None
    for _ in range(2):
        t0 = time.perf_counter()
        r = ctx.count_device(dp, nb, do, dl, NR)
        ts.append(time.perf_counter() - t0)
        info = dict(r.info)
        del r
    ctx.synth_free(dp, do, dl)
    ctx.close()
    nk = NR * (READ_LEN - K + 1)
    dt = sum(ts) / len(ts)
    return {"name": "configs2_input_on_one_gpu", "what": "%.0f Gbp of reads (BASELINE configs[2]'s whole input) in one hsk_count_device call on one GPU" % gbp, "bases": NR * READ_LEN,
            "packed_bytes": int(nb), "kmers": nk, "value": nk / dt, "unit": "k-mers/s", "ms_per_call": dt * 1e3, "entries": info.get("n"), "ntasks": info["ntasks"],
            "phases_ms": {k_: round(v, 2) for k_, v in info.items() if k_.startswith("ms_")}}


def first_call_leg(H, local, genome_len, nreads, seed):
    """Does a call depend on what the context counted before?  ONE context (pools grown once per kind of input, untimed), then for every
    kind: a call on ANOTHER kind (error-free reads; uniform reads for the error-free kind), the call on this kind right after it = `after_other_input_ms`
    (the context's memory points the wrong way), and two more = `steady_ms`.  hysortk::kmer_count() is called once per process (reference
    src/hysortk.cpp:36-96): the two must agree."""
    ctx = H.Context(K=K, M=M, L=1, U=65535, device=local, keep_device=True, profile=True)
    kinds = [("error_free", 0.0), ("errors_0.3pct", 0.003), ("errors_1pct", 0.01), ("uniform", 0.75)]
    data = {name: ctx.synth_reads(genome_len, READ_LEN, nreads, seed + i, error_rate=er) for i, (name, er) in enumerate(kinds)}
    nk = nreads * (READ_LEN - K + 1)

    def call(name):
        dp, nb, do, dl = data[name]
        ctx.stats(reset=True)
        t0 = time.perf_counter()
        r = ctx.count_device(dp, nb, do, dl, nreads)
        ms = (time.perf_counter() - t0) * 1e3
        st = ctx.stats(reset=True)
        dev = r.info["ms_total"]
        del r
        return ms, dev, int(st.get("combine_launches", 0)), int(st.get("hist_launches", 0))
    for name, _ in kinds:                                  # pools
        call(name)
    out = []
    for name, er in kinds:
        other = "uniform" if name == "error_free" else "error_free"
        call(other)
        a = call(name)
        b = [call(name) for _ in range(2)]
        steady = sum(x[0] for x in b) / len(b)
        out.append({"input": name, "error_rate": er, "kmers": nk, "after_other_input_ms": a[0], "steady_ms": steady, "ratio": a[0] / steady,
                    "after_other_input_device_ms": a[1], "steady_device_ms": sum(x[1] for x in b) / len(b),
                    "combine_launches": {"after_other_input": a[2], "steady": b[-1][2]}, "instance_extraction_launches": {"after_other_input": a[3], "steady": b[-1][3]}})
    for d in data.values():
        ctx.synth_free(d[0], d[2], d[3])
    ctx.close()
    return out


def multi_rank_leg(H, local, name, KK, ext, R, per_rank_bp, ntasks, seed, steps=2):
    """The N > 1 data path on ONE GPU at scale (hsk_count_loopback_device): R virtual ranks, each holding per_rank_bp of reads sampled from one
    genome of R x per_rank_bp / 32 bases (the 8-GPU workload's shape: (R-1)/R of the supermers change ranks), the 8-GPU bench's task count;
    task-size probe, dispatcher, owner-grouped byte-store placement, grouped exchange overlapped with the sort (device copies stand in for RCCL
    send / recv over xGMI), multi-segment extraction.  The virtual ranks run one after the other, so a rank's device ms IS what one GPU of the
    8-GPU job spends outside the wire: the per-GPU cost of leaving the single-GPU plan, driver-timed."""
    G = R * per_rank_bp // COVERAGE
    NR = per_rank_bp // READ_LEN
    ctx = H.Context(K=KK, M=M, L=L, U=U, EXT=ext, ntasks=ntasks, device=local, profile=True, keep_device=True)
    reads = []
    for r in range(R):
        dp, nb, do, dl = ctx.synth_reads(G, READ_LEN, NR, seed, first_read=r * NR)
        reads.append((dp, nb, do, dl, NR))
    res, owner = ctx.count_loopback_device(reads)          # warm-up: pools
    del res
    ctx.stats(reset=True)
    walls, ranks = [], None
    for _ in range(steps):
        t0 = time.perf_counter()
        res, owner = ctx.count_loopback_device(reads)
        walls.append(time.perf_counter() - t0)
        ranks = [dict(r_.info) for r_ in res]
        del res
    st = ctx.stats(reset=True)
    for x in reads:
        ctx.synth_free(x[0], x[2], x[3])
    ctx.close()
    wall = sum(walls) / len(walls)
    nk_rank = NR * (READ_LEN - KK + 1)
    per = [{"rank": i, "owned_kmers": r_["total_kmers"], "owned_tasks": int((owner == i).sum()),
            "ms": {k_[3:]: round(v, 2) for k_, v in r_.items() if k_.startswith("ms_") and k_ != "ms_d2h"}} for i, r_ in enumerate(ranks)]
    ms_rank = [p_["ms"]["total"] for p_ in per]
    mean_ms = sum(ms_rank) / R
    return {"name": name, "K": KK, "EXT": ext, "virtual_ranks": R, "bp_per_rank": per_rank_bp, "kmers_per_rank": nk_rank, "ntasks": len(owner), "steps": steps,
            "wall_ms_all_ranks": wall * 1e3, "value": R * nk_rank / wall, "unit": "k-mers/s (all virtual ranks on one GPU, one after the other)",
            "per_rank_device_ms": {"mean": mean_ms, "max": max(ms_rank), "min": min(ms_rank)},
            "per_gpu_rate_from_device_ms": nk_rank / (max(ms_rank) * 1e-3),
            "per_gpu_rate_note": "k-mers of one rank's reads / the slowest rank's device ms: what one GPU of the 8-GPU job would sustain if the wire were free "
                                 "(the exchange here is device copies inside one HBM; over xGMI a rank moves ~1.55 B per k-mer x 7/8 to its peers on seven links)",
            "ranks": per, "combine_launches": int(st.get("combine_launches", 0)), "exchange": "device-to-device copies in place of ncclSend / ncclRecv (hsk_comm.h plans, unchanged)"}


def e2e_host_leg(H, KK, ext, ntasks, local, genome_len, nreads, seed, steps):
    """hsk_count() from a DnaBuffer in pinned host RAM to KmerListS entries in pinned host RAM (what the reference times as
    'Overall kmer counting (Excluding I/O)', src/hysortk.cpp:58,91), L=15 U=40."""
    ctx = H.Context(K=KK, M=M, L=L, U=U, EXT=ext, ntasks=ntasks, device=local, profile=True, keep_device=False)
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed)
    packed = H.pinned_empty(nb, np.uint8)
    off = H.pinned_empty(nreads, np.uint64)
    lens = H.pinned_empty(nreads, np.uint32)
    ctx.d2h_into(packed, dp, nb)
    ctx.d2h_into(off, do, nreads * 8)
    ctx.d2h_into(lens, dl, nreads * 4)
    ctx.synth_free(dp, do, dl)
    ctx.count_host_timed(packed, off, lens)                   # warm-up: pools, pinned result block, entries-per-k-mer estimate
    ctx.stats(reset=True)
    secs, n_entries, info = [], 0, None
    for _ in range(steps):
        s, n_entries, info = ctx.count_host_timed(packed, off, lens)
        secs.append(s)
    st = ctx.stats(reset=True)
    ctx.close()
    H.pinned_free(packed); H.pinned_free(off); H.pinned_free(lens)
    nk = nreads * (READ_LEN - KK + 1)
    mean = sum(secs) / len(secs)
    return {"value": nk / mean, "unit": "k-mers/s", "ms_per_step": mean * 1e3, "steps": steps, "L": L, "U": U, "entries": n_entries,
            "input": "DnaBuffer in pinned host memory (hsk_host_alloc): the packed reads cross PCIe as 16 DMA slabs pipelined with the minimizer scan and the placement; "
                     "of the read index nothing travels for fixed-length reads (lengths generated on the device from a sample, offsets derived from them; host threads verify "
                     "every length and offset while the GPU scans)",
            "output": "KmerListS entries + histogram in pinned host memory; a batch's entries cross PCIe as 7 bytes each (low 48 key bits + 8-bit count below a per-task prefix "
                      "directory) while later batches are counted, and are widened to the 16-byte KmerListEntryS layout by host threads",
            "h2d_bytes_per_step": st["h2d_bytes"] / steps, "d2h_bytes_per_step": st["d2h_bytes"] / steps,
            "h2d_ms": st["h2d_ms"] / steps, "d2h_ms": st["d2h_ms"] / steps,
            "device_ms_total": info["ms_total"], "host_syncs_per_step": st["host_syncs"] / steps, "host_waits_covered_per_step": st["host_waits_covered"] / steps}


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
    dp, nb, do, dl = ctx.synth_reads(genome_len, READ_LEN, nreads, seed, first_read=rank * nreads, error_rate=a.error_rate)

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    info = None
    first_call = None
    for w_ in range(a.warmup):
        t_w = time.perf_counter()
        info = ctx.count_device(dp, nb, do, dl, nreads, rid_base=rank * nreads).info
        if w_ == 0:                                          # the very first call of this process and context: what a one-shot client pays
            first_call = {"wall_ms": (time.perf_counter() - t_w) * 1e3, "device_ms": info["ms_total"],
                          "note": "first hsk_count_device of a fresh process (the --warmup call): the plan is chosen inside the call, so everything beyond the steady "
                                  "state is the runtime mapping the pools' device memory for the first time (hipMalloc: ~20-60 ms per GB on this stack, tools/exp/malloc_cost.hip) "
                                  "and loading the code objects"}
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
        S = a.steps
        total_kmers = nk_rank * world * S
        value = total_kmers / dt
        ms_step = dt / S * 1e3
        launches = max(int(st["scatter_launches"]), 1)
        avg_ms = st["scatter_ms"] / launches
        bytes_per_launch = st["scatter_bytes"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # PMC evidence of this build (tools/gpu_profile_round.sh -> profiles/pmc.json: per kernel and step, HBM bytes and instruction counts)
        # and the VALU issue cost of the VALU-bound kernels' instruction mix (tools/valu_floor.py -> profiles/valu_mix.json, from
        # profiles/r04_valu_rates.txt).  Nothing below is a literal: every count comes from those files.
        import hashlib

        def load_json(name):
            try:
                return json.load(open(os.path.join(ROOT, "profiles", name)))
            except Exception:
                return {}
        pmc, mix = load_json("pmc.json"), load_json("valu_mix.json")
        try:
            from hysortk_amd.build import source_sha16
            lib_sha = source_sha16()                                      # (the sources the running library was built from)
        except Exception:
            lib_sha = None
        pmc_kmers = (pmc.get("workload") or {}).get("kmers_per_step")
        pmc_scale = (nk_rank / pmc_kmers) if pmc_kmers else 1.0           # (a --scale run: the counts scale with the workload)
        pmc_src = {"file": "profiles/pmc.json", "produced_by": "tools/gpu_profile_round.sh (rocprofv3 --pmc, one counter set per pass, kernel trace only)",
                   "build_sha16_of_passes": pmc.get("build_sha16"), "build_sha16_running": lib_sha, "same_build": bool(lib_sha) and pmc.get("build_sha16") == lib_sha,
                   "scaled_by": pmc_scale, "measured_in_this_run": False}

        def pmc_get(prefix, field):
            vals = [d[field] for name, d in (pmc.get("kernels") or {}).items() if name.startswith(prefix) and field in d]
            return sum(vals) * pmc_scale if vals else None
        path_traffic = (pmc.get("path_hbm_bytes_per_step") or 0) * pmc_scale or None
        rec = 8 * ((KK + 31) // 32)
        copy_peak = ctx.copy_peak(1 << 32, 3) if world == 1 else None      # hand-written 16-byte-per-lane copy (hsk_copy_peak), measured in this run
        NSIMD, CLK = 256 * 4, 2.4e9

        def kern(name, ms, n, alg_bytes, bound, note, pmc_prefix=None, mix_key=None):
            d = {"kernel": name, "ms_per_step": ms / S, "launches_per_step": n / S, "avg_launch_ms": (ms / n) if n else None, "bound": bound, "note": note}
            if alg_bytes and ms > 0:
                d["algorithmic_bytes_per_launch"] = alg_bytes / max(n, 1)
                d["algorithmic_GBs"] = alg_bytes / (ms * 1e-3) / 1e9
                d["frac_of_hbm_peak"] = d["algorithmic_GBs"] / HBM_PEAK_GBS
                if copy_peak:
                    d["frac_of_copy_peak"] = d["algorithmic_GBs"] / copy_peak
            hb = pmc_get(pmc_prefix or name, "hbm_bytes_per_step")
            d["pmc_traffic_bytes_per_launch"] = (hb / (n / S)) if hb and n else None
            if hb and ms > 0:
                d["pmc_traffic_GBs"] = hb / (ms / S * 1e-3) / 1e9
            vi = pmc_get(pmc_prefix or name, "SQ_INSTS_VALU_per_step")
            mk = (mix.get("kernels") or {}).get(mix_key or "")
            if bound == "valu" and vi and mk and ms > 0:
                cyc = mk["avg_cycles_per_valu_wave_instruction"]
                floor_ms = vi * cyc / (NSIMD * CLK) * 1e3
                d["valu"] = {"wave_instructions_per_step": vi, "lane_ops_per_s": vi * 64 / (ms / S * 1e-3), "avg_issue_cycles_per_wave_instruction": cyc,
                             "mix_ceiling_lane_ops_per_s": 64 * NSIMD * CLK / cyc, "floor_ms_per_step": floor_ms, "frac_of_issue_floor": floor_ms / (ms / S),
                             "salu_wave_instructions_per_step": pmc_get(pmc_prefix or name, "SQ_INSTS_SALU_per_step"), "lds_wave_instructions_per_step": pmc_get(pmc_prefix or name, "SQ_INSTS_LDS_per_step"),
                             "how": "floor = SQ_INSTS_VALU (profiles/pmc.json) x the average issue cycles of the kernel's static instruction mix (profiles/valu_mix.json: every mnemonic "
                                    "priced with profiles/r04_valu_rates.txt at 4 waves per SIMD) / (1024 SIMDs x 2.4 GHz).  Measured rates: v_mov / v_and / v_or / v_sub / v_add / v_xor / v_not / "
                                    "v_lshrrev_b32 / v_bitop3 issue in 2.3 - 2.8 cycles per wave-instruction (the SIMD-32 rate: 0.72 - 0.84 of 78.6 T lane-ops/s); every multiply, 64-bit "
                                    "operation, left shift, three-operand add / logic, compare, select, DPP and cross-lane instruction in 4.1 - 5.2 cycles (31 - 38 T lane-ops/s)"}
            return d
        nsup = st["place_supermers"] or (info.get("total_supermers", 0) * S)      # (scan-placed items: no placement launch to count them)
        item_mode = bool(st.get("combine_launches"))
        scan_places = item_mode and not st["place_launches"]                        # round 4: scan_kernel writes the 16-byte items itself (ParseArgs::bin_*)
        kernels = [
            kern("scan_kernel", st["scan_ms"], st["scan_launches"], st["scan_bytes"] + nsup * (20.0 if scan_places else 8.0 if item_mode else 4.0), "valu",
                 "minimizer hashes + supermer cuts: bound by VALU issue (MurmurHash3 of every m-mer: six 64-bit multiplies and the xor-shifts, then window minima and supermer cuts); "
                 "algorithmic bytes = packed reads in + per supermer out: a 4-byte record (instance path), 8 bytes of record + minimizer bits (combining extraction, the default: place_items_kernel "
                 "makes the items), or the 16-byte item + 4 bytes of minimizer bits when the scan places them itself (tuning scan_place=1, measured break-even: DESIGN.md 3.2h)" +
                 (" -- this run: the scan placed the items" if scan_places else "") + "; its HBM fraction says nothing about it",
                 pmc_prefix="scan_kernel", mix_key="scan_kernelILi%dELi17E" % KK if KK in (31, 51) else "scan_kernelILi31ELi17E"),
            kern("expand_scatter2_kernel" if KK <= 32 and not a.ext else "expand_scatter_kernel", st["hist_ms"], st["hist_launches"], st["hist_bytes"] * (1 + 1.1 / rec), "hbm",
                 "k-mer extraction fused with the first scatter pass: reads the supermers (1.1 B per k-mer), writes the keys into chunk-listed digit bins; "
                 "a chain of short phases per 3700-key flush, no unit of the CU saturated: bound by how many workgroups a CU holds (three since round 3: the two-sweep kernel)", pmc_prefix="expand_scatter"),
            kern("onesweep_multi_kernel", st["scatter_ms"], st["scatter_launches"], st["scatter_bytes"], "hbm", "second radix scatter pass over chunk tiles: 2 x record bytes per key"),
            kern("agg_finish_kernel", st["agg_ms"], st["agg_launches"], st["agg_bytes"] + (info.get("n", 0) or 0) * (rec + 8.0) * S, "lds-issue",
                 "per-prefix-bin LDS hash aggregation: reads every record once (with the combining extraction: 16-byte {k-mer, count} pairs), writes the kept entries (16 B each); bound by "
                 "instruction issue around the LDS probes (scalar unit + LDS queue; the probe loop is hand-written assembly for that reason), not by HBM"),
            kern("place_kernel", st["place_ms"], st["place_launches"], nsup * 13.0, "hbm-scattered",
                 "supermers to their task slots: 4-byte records in, 9 bytes per supermer out in short runs"),
        ]
        if item_mode:
            # the combining extraction ran (hsk_combine.h): no instance extraction, the scatter pass and the finish work on {k-mer, count} pairs.
            # Per supermer item (16 bytes + 4 of minimizer bits): the placement reads its 4 + 4 bytes of records and ~2.4 bytes of packed bases and
            # writes 20; the bucket order reads 4 (histogram), then 4 + 16 and writes 16.
            kernels = [k_ for k_ in kernels if k_["kernel"] not in ("expand_scatter2_kernel", "expand_scatter_kernel", "place_kernel")]
            items = st["bucket_items"]
            kernels += ([] if scan_places else [
                kern("place_items_kernel", st["place_ms"], st["place_launches"], nsup * 30.4, "hbm-scattered",
                     "supermers (as 16-byte items: 64 bases + k-mer count, with 4 bytes of minimizer bits) to their virtual-task slots (16 virtual tasks per task: the top "
                     "minimizer bits); bases read once from the packed reads staged in LDS; 8 bytes of records in, 20 out in runs of ~25 items")]) + [
                kern("bucket_scatter_kernel", st["bucket_ms"], st["bucket_launches"], items * 40.0, "hbm-scattered",
                     "bucket order of the items inside a virtual task (histogram launch + scatter launch): 8192 items staged and ordered in LDS, every bucket's run written in one piece (~8 items = 128 bytes)",
                     pmc_prefix="bucket_"),
                kern("combine_kernel", st["combine_ms"], st["combine_launches"], items * 16.0 + st["combine_pairs"] * 16.0, "valu",
                     "extraction + counting per minimizer bucket: reads the bucket's items (16 B per supermer), rolls the k-mers into a 2048-slot LDS hash table, writes the table's "
                     "{k-mer, count} pairs into the digit bins of the first radix pass; bound by VALU issue, not by HBM or LDS latency -- an item is a supermer of 7.6 k-mers on average in "
                     "16 slots, so the waves work at ~47 %% of their lanes (%.1f k-mers per pair in this run)" % (st["combine_kmers"] / max(st["combine_pairs"], 1)),
                     pmc_prefix="combine_kernel", mix_key="combine_kernelILi31E"),
            ]
        kernels.sort(key=lambda d: -d["ms_per_step"])
        # The roofline block names the kernel with the MOST TIME, whatever bounds it, with its own ceiling (a VALU-bound kernel: the issue floor above); the
        # time-dominant HBM-bound kernel stands beside it (`hbm_leader`) with the block's usual fields.
        dom = next(d for d in kernels if d["bound"].startswith("hbm"))
        lead = kernels[0]

        def hbm_block(d):
            return {"bound": "hbm", "kernel": d["kernel"], "ms_per_step": d["ms_per_step"], "achieved": d.get("algorithmic_GBs", 0.0), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d.get("frac_of_hbm_peak", 0.0), "traffic": d["pmc_traffic_bytes_per_launch"], "launches": int(d["launches_per_step"] * S),
                    "avg_launch_ms": d["avg_launch_ms"], "bytes_per_launch": d.get("algorithmic_bytes_per_launch"), "frac_of_copy_peak": d.get("frac_of_copy_peak")}
        ms_total = phase.get("ms_total", 0) / S
        ref_b = 152.3 if KK <= 32 and not a.ext else (464.4 if KK > 32 else 304.3)
        dev = {"value": value, "unit": "k-mers/s", "ms_per_step": ms_step, "input": "resident in HBM", "output": "left in HBM"}
        out = {
            "metric": "k-mers counted/sec at K=%d" % KK, "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": S, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_is": "device_resident (inputs already in HBM when the timed region starts, as the bench contract asks); `host_to_host` in this line is the same "
                        "workload through hsk_count() from host memory to host memory = the metric as SURVEY 8(d) and the reference's timer define it",
            "dtype": ("u64" if KK <= 32 else "u128") + ("" if not a.ext else "+u64 payload"), "data": "synthetic",
            "config": {"workload": "S-reads(G=%d bp x %d GPU, c=%d): %d x %d-bp reads per GPU = %.3g bp, %d k-mers per GPU" % (
                int(GENOME_PER_GPU * a.scale), world, COVERAGE, nreads, READ_LEN, nreads * READ_LEN, nk_rank),
                "K": KK, "M": M, "L": L, "U": U, "EXT": a.ext, "ntasks": info["ntasks"], "scale": a.scale, "error_rate": a.error_rate,
                "input": "resident in HBM", "output": "left in HBM (entries=%d on rank 0)" % info.get("n", -1),
                "exchange": "RCCL send/recv all-to-all-v" if world > 1 else "none"},
            "device_resident": dev, "first_call_in_process": first_call,
            "roofline": {**({"bound": "valu", "kernel": lead["kernel"], "ms_per_step": lead["ms_per_step"], "of_ms_per_step": ms_total,
                             "achieved": lead["valu"]["lane_ops_per_s"] / 1e12, "peak": lead["valu"]["mix_ceiling_lane_ops_per_s"] / 1e12, "unit": "T lane-op/s (integer VALU issue; no MFMA on this path)",
                             "frac": lead["valu"]["frac_of_issue_floor"], "floor_ms": lead["valu"]["floor_ms_per_step"], "traffic": lead["pmc_traffic_bytes_per_launch"],
                             "launches": int(lead["launches_per_step"] * S), "avg_launch_ms": lead["avg_launch_ms"], "valu": lead["valu"],
                             "as_hbm": {"achieved": lead.get("algorithmic_GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lead.get("frac_of_hbm_peak"),
                                        "pmc_GBs": lead.get("pmc_traffic_GBs"), "note": "the same kernel against the HBM roofline: it is not what bounds it"}}
                            if lead.get("valu") else hbm_block(lead)),
                         "time_dominant": "%s: %.1f of %.1f ms per step" % (lead["kernel"], lead["ms_per_step"], ms_total),
                         "hbm_leader": hbm_block(dom),
                         "traffic_source": pmc_src,
                         "measured_copy_peak_GBs": copy_peak, "measured_copy_peak_source": "hsk_copy_peak in this run: hand-written kernel, 16 B per lane, 4 GiB read + 4 GiB written, best of 12 launches",
                         "whole_path": {
                             "traffic_bytes_per_step": path_traffic, "traffic_source": "profiles/pmc.json (see roofline.traffic_source)",
                             "GBs": (path_traffic / (ms_total * 1e-3) / 1e9) if path_traffic and ms_total else None,
                             "frac_of_hbm_peak": (path_traffic / (ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS) if path_traffic and ms_total else None,
                             "frac_of_copy_peak": (path_traffic / (ms_total * 1e-3) / 1e9 / copy_peak) if path_traffic and ms_total and copy_peak else None,
                             "reference_algorithm_GBs": (ref_b * nk_rank) / (ms_total * 1e-3) / 1e9 if ms_total else None,
                             "reference_algorithm_note": "SURVEY 8(d) fixed accounting (8-bit LSD over all key bytes: 152.3 B per 31-mer) x k-mers / device time: the REFERENCE "
                                                         "algorithm's byte count, not bytes this build moves (2 passes + LDS aggregation, ~49 B per k-mer) and so not a roofline fraction; "
                                                         "variants[full_sort] runs that algorithm for real"}},
            "kernels": kernels,
            "plan": ("combining extraction: supermers ordered by minimizer bucket, k-mers counted in LDS tables where they are extracted, %.1f k-mers per {k-mer, count} pair into one "
                     "scatter pass + weighted LDS finish (hsk_combine.h; HSK_FLAG_NO_COMBINE / variants[instance_path]: every instance through two passes)" % (st["combine_kmers"] / max(st["combine_pairs"], 1)))
                    if st.get("combine_launches") else "instance path: every k-mer instance through two radix passes + LDS aggregation",
            "host_syncs_per_step": st["host_syncs"] / S, "host_waits_covered_per_step": st["host_waits_covered"] / S,
            "path_stats": {k_: int(st[k_]) for k_ in ("fused_tasks", "redone_tasks", "agg_retried_tasks", "parse_fallbacks", "heavy_tasks") if k_ in st},
            "phases_ms_per_step": {k_: v / S for k_, v in sorted(phase.items())},
        }
        ctx.synth_free(dp, do, dl)
        ctx.close()
        if world == 1 and not a.no_e2e:
            try:
                out["host_to_host"] = e2e_host_leg(H, KK, a.ext, a.ntasks, local, genome_len, nreads, seed, max(a.steps, 3))
            except Exception as e:
                out["host_to_host"] = {"error": str(e)[:300]}
        only = set(x for x in a.legs.split(",") if x)
        if world == 1 and not a.no_variants:
            G = int(GENOME_PER_GPU * a.scale)
            legs = [
                ("k51", "BASELINE configs[3]'s record shape (two-word keys) on the same reads", 51, 0, None, L, U, G, nreads, 0.0, 464.4),
                ("ext", "BASELINE configs[4]'s record shape (EXTENSION=1: pos + rid carried through the sort and grouped per k-mer)", 31, 1, None, L, U, G, nreads, 0.0, 304.3),
                ("instance_path", "the headline workload with HSK_FLAG_NO_COMBINE: every k-mer instance extracted and moved through two radix passes, then the LDS aggregation "
                                  "(the default plan of rounds 1-2 and of every record shape the combining extraction does not take)", 31, 0, "no_combine", L, U, G, nreads, 0.0, 152.3),
                ("no_aggregation", "the headline workload with HSK_FLAG_NO_AGGREGATION: four scatter passes on the top 32 bits + in-LDS tile finish", 31, 0, "no_aggregation", L, U, G, nreads, 0.0, 152.3),
                ("full_sort", "the headline workload with HSK_FLAG_FULL_SORT = the algorithm north_star names: LSD radix sort over all 8 key bytes "
                              "(reference sort_task, src/kmerops.cpp:1383) + adjacent-equal merge-count (count_sorted_kmers, :1410)", 31, 0, "full_sort", L, U, G, nreads, 0.0, 152.3),
                ("errors_0.3pct", "the headline workload with 0.3 % substitution errors per base (instance path: one k-mer in eleven is an error k-mer)", 31, 0, None, L, U, G, nreads, 0.003, 152.3),
                ("errors_1pct", "the headline workload with 1 % substitution errors per base (a realistic short-read error rate; instance path)", 31, 0, None, L, U, G, nreads, 0.01, 152.3),
                ("uniform", "uniform random reads (error rate 0.75 makes every base uniform: all counts 1), L=1 U=65535: SURVEY 8(d)'s worst case for output volume, "
                            "half the headline's bases so that the 16 B per k-mer of output stay in HBM", 31, 0, None, 1, 65535, G // 2, nreads // 2, 0.75, 152.3),
            ]
            out["variants"] = []
            for (name, note, vk, vext, plan, Lv, Uv, vg, vn, er, refb) in legs:
                if only and name not in only:
                    continue
                try:
                    out["variants"].append(run_variant(H, local, name, note, vk, vext, plan, Lv, Uv, vg, vn, seed, er, 3, refb))
                except Exception as e:
                    out["variants"].append({"name": name, "error": str(e)[:300]})
        if world == 1 and not a.no_variants and (not only or "multi_rank" in only):
            out["multi_rank_path"] = []
            for (mname, mk, mext, mbp) in (("k31_8x5Gbp", 31, 0, 5_000_000_000), ("k51_8x2.5Gbp", 51, 0, 2_500_000_000), ("ext_8x1.25Gbp", 31, 1, 1_250_000_000)):
                try:
                    out["multi_rank_path"].append(multi_rank_leg(H, local, mname, mk, mext, 8, int(mbp * a.scale), 320, seed + 7))
                except Exception as e:
                    out["multi_rank_path"].append({"name": mname, "error": str(e)[:300]})
        if world == 1 and not a.no_variants and a.scale == 1.0 and (not only or "large" in only):
            try:
                out["large_input"] = large_input_leg(H, local, 80.0, seed + 200)
            except Exception as e:
                out["large_input"] = {"error": str(e)[:300]}
        if world == 1 and not a.no_variants and (not only or "first_calls" in only):
            try:
                out["first_calls"] = first_call_leg(H, local, int(GENOME_PER_GPU * a.scale) // 2, nreads // 2, seed + 100)
            except Exception as e:
                out["first_calls"] = {"error": str(e)[:300]}
        if world == 1 and not a.no_cpu:
            ncores = os.cpu_count() or 1
            div = max(a.cpu_div, 1)
            g_s = max(int(GENOME_PER_GPU * a.scale) // div, 10000)
            try:
                out["cpu_baseline"] = cpu_baseline(H, local, g_s, g_s * COVERAGE // READ_LEN, seed + 1, ncores, 1.0 / div)
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)[:300]}
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        ctx.synth_free(dp, do, dl)
        ctx.close()
    if comm is not None:
        comm.barrier()
        comm.destroy()


if __name__ == "__main__":
    main()
