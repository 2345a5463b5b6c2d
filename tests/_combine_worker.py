"""Helper of tests/test_gpu_combine.py (run_spec, called in process; `python _combine_worker.py <json>` prints the same as JSON lines): counts one
synthetic input on a context with the tuning the test chose and returns one dict per call: digest of the list, entries, and the
statistics that tell which plan ran.  spec: K, M, L, U, ntasks, genome, read_len, nreads, seed, error_rate, calls (list of
"device" | "host" | "pinned"), plan."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hysortk_amd as H  # noqa: E402


def digest(r):
    h = hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes())
    if r.pos is not None and len(r.cnt):               # EXTENSION: the payloads of a k-mer come in any order, and the arrays may hold the payloads of filtered
        # k-mers between the slices of the kept ones: an order-independent sum over every kept k-mer's own slice [payload_off[i], + cnt[i])
        x = r.pos.astype(np.uint64) | (r.rid.astype(np.uint64) << np.uint64(32))
        with np.errstate(over="ignore"):
            cs = np.concatenate((np.zeros(1, np.uint64), np.cumsum(x * np.uint64(0x9E3779B97F4A7C15), dtype=np.uint64)))
            o = r.payload_off[:len(r.cnt)].astype(np.int64)
            h.update((cs[o + r.cnt.astype(np.int64)] - cs[o]).tobytes())
    return h.hexdigest()


def run_spec(spec):
    """counts one synthetic input on a context of its own; returns one dict per call (tests/test_gpu_combine.py calls this in process)"""
    out = []
    ctx = H.Context(K=spec["K"], M=spec["M"], L=spec["L"], U=spec["U"], EXT=spec.get("EXT", 0), ntasks=spec["ntasks"], profile=True, plan=spec.get("plan"), tuning=spec.get("tuning"))
    dp, nb, do, dl = ctx.synth_reads(spec["genome"], spec["read_len"], spec["nreads"], spec["seed"], error_rate=spec.get("error_rate", 0.0))
    n = spec["nreads"]
    packed = np.empty(nb, np.uint8); off = np.empty(n, np.uint64); lens = np.empty(n, np.uint32)
    ctx.d2h_into(packed, dp, nb); ctx.d2h_into(off, do, n * 8); ctx.d2h_into(lens, dl, n * 4)
    if spec.get("poly_a_pct"):
        # a share of the reads replaced by all-A reads: ONE k-mer, ONE minimizer bucket for all of them ("host" / "pinned" calls only)
        rng = np.random.default_rng(spec["seed"] + 1)
        view = packed.reshape(n, (spec["read_len"] + 3) // 4)
        view[rng.choice(n, int(n * spec["poly_a_pct"] / 100), replace=False)] = 0
        assert all(h in ("host", "pinned") or h.startswith("hostloop:") for h in spec["calls"])
    pinned = None
    for how in spec["calls"]:
        ctx.stats(reset=True)
        if how.startswith("loopback:") or how.startswith("hostloop:"):
            # R virtual ranks over the same reads, split evenly (hsk_count_loopback_device): one line per call, digest over the ranks' lists in rank order
            R = int(how.split(":")[1])
            per = n // R
            nbr = (spec["read_len"] + 3) // 4
            if how.startswith("hostloop:"):                # the host's arrays (with the all-A reads, if any): hsk_count_loopback
                res, owner = ctx.count_loopback([(packed[r * per * nbr:(r + 1) * per * nbr], off[:per], lens[:per]) for r in range(R)])
            else:
                reads = [(dp.value + r * per * nbr, per * nbr, do, dl, per) for r in range(R)]      # (fixed-length reads: every rank's offsets start at 0 again -- the same array serves)
                res, owner = ctx.count_loopback_device(reads)
            st = ctx.stats(reset=True)
            h = hashlib.sha256()
            ent = 0
            for t in range(len(owner)):
                kl = res[int(owner[t])]
                a, b = int(kl.task_off[t]), int(kl.task_off[t + 1])
                h.update(kl.kmers[a:b].tobytes()); h.update(kl.cnt[a:b].tobytes()); ent += b - a
            out.append({"how": how, "digest": h.hexdigest(), "entries": ent, "total_kmers": int(sum(k.info["total_kmers"] for k in res)),
                              "combine_launches": int(st["combine_launches"]), "combine_pairs": int(st["combine_pairs"]), "combine_kmers": int(st["combine_kmers"]),
                              "instance_extractions": int(st["hist_launches"]), "fused_tasks": int(st["fused_tasks"]), "redone_tasks": int(st["redone_tasks"]),
                              "dropped_kmers": int(st.get("dropped_kmers", 0))})
            continue
        if how == "device":
            r = ctx.count_device(dp, nb, do, dl, n)
        elif how == "host":
            r = ctx.count((packed, off, lens))
        else:
            if pinned is None:
                pinned = (H.pinned_empty(nb, np.uint8), H.pinned_empty(n, np.uint64), H.pinned_empty(n, np.uint32))
                pinned[0][:] = packed; pinned[1][:] = off; pinned[2][:] = lens
            r = ctx.count(pinned)
        st = ctx.stats(reset=True)
        out.append({"how": how, "digest": digest(r), "entries": len(r), "total_kmers": int(r.info["total_kmers"]),
                          "combine_launches": int(st["combine_launches"]), "combine_pairs": int(st["combine_pairs"]), "combine_kmers": int(st["combine_kmers"]),
                          "instance_extractions": int(st["hist_launches"]), "fused_tasks": int(st["fused_tasks"]), "redone_tasks": int(st["redone_tasks"]),
                          "dropped_kmers": int(st.get("dropped_kmers", 0))})
    if spec.get("dump"):
        np.savez(spec["dump"], kmers=r.kmers, cnt=r.cnt, task_off=r.task_off, packed=packed, off=off, lens=lens)
    if pinned is not None:
        for a in pinned:
            H.pinned_free(a)
    ctx.synth_free(dp, do, dl)
    ctx.close()
    return out


def main():
    for d in run_spec(json.loads(sys.argv[1])):
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
