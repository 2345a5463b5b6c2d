"""Worker of tests/test_gpu_combine.py: counts one synthetic input under the environment the parent chose (the switches of the
combining extraction are read once per process) and prints one JSON line per call: digest of the list, entries, and the
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
    return hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes()).hexdigest()


def main():
    spec = json.loads(sys.argv[1])
    ctx = H.Context(K=spec["K"], M=spec["M"], L=spec["L"], U=spec["U"], ntasks=spec["ntasks"], profile=True, plan=spec.get("plan"))
    dp, nb, do, dl = ctx.synth_reads(spec["genome"], spec["read_len"], spec["nreads"], spec["seed"], error_rate=spec.get("error_rate", 0.0))
    n = spec["nreads"]
    packed = np.empty(nb, np.uint8); off = np.empty(n, np.uint64); lens = np.empty(n, np.uint32)
    ctx.d2h_into(packed, dp, nb); ctx.d2h_into(off, do, n * 8); ctx.d2h_into(lens, dl, n * 4)
    pinned = None
    for how in spec["calls"]:
        ctx.stats(reset=True)
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
        print(json.dumps({"how": how, "digest": digest(r), "entries": len(r), "total_kmers": int(r.info["total_kmers"]),
                          "combine_launches": int(st["combine_launches"]), "combine_pairs": int(st["combine_pairs"]), "combine_kmers": int(st["combine_kmers"]),
                          "instance_extractions": int(st["hist_launches"]), "fused_tasks": int(st["fused_tasks"])}), flush=True)
    if spec.get("dump"):
        np.savez(spec["dump"], kmers=r.kmers, cnt=r.cnt, task_off=r.task_off, packed=packed, off=off, lens=lens)
    if pinned is not None:
        for a in pinned:
            H.pinned_free(a)
    ctx.synth_free(dp, do, dl)
    ctx.close()


if __name__ == "__main__":
    main()
