#!/usr/bin/env python3
"""VALU issue cost of the VALU-bound kernels from their gfx950 ISA and the measured issue rates.

  python3 tools/valu_floor.py profiles/r04_valu_rates.txt profiles/valu_mix.json [kernel-name-substring ...]

For every named kernel (default: the scan and the combining extraction): the static VALU mix of its code object (hipcc -S),
every mnemonic priced with the cycles per wave-instruction per SIMD that tools/exp/mulrate measured for its class at four
waves per SIMD (the occupancy these kernels run at), and the mix's average.  bench.py multiplies that average with the
DYNAMIC count of VALU wave-instructions (SQ_INSTS_VALU of the PMC pass, profiles/pmc.json) to get the kernel's issue floor:

    floor_ms = SQ_INSTS_VALU x avg_cycles / (1024 SIMDs x 2.4 GHz)

The approximation is the static mix standing in for the dynamic one (the hot loops of these kernels are most of their code).
No GPU needed: runs where hipcc cross-compiles."""
import json, os, re, subprocess, sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hysortk_amd", "csrc", "hsk_api.hip")

# mnemonic (without _e32 / _e64 / _dpp / _sdwa suffix) -> class key of tools/exp/mulrate.hip
CLASS = {
    "v_mov_b32": "mov", "v_and_b32": "and_or32", "v_or_b32": "and_or32", "v_xor_b32": "add32", "v_add_u32": "add32", "v_sub_u32": "sub32", "v_subrev_u32": "sub32",
    "v_not_b32": "not", "v_lshrrev_b32": "shr32", "v_lshlrev_b32": "shl32", "v_ashrrev_i32": "shr32", "v_bitop3_b32": "bitop3",
    "v_add3_u32": "add3", "v_mul_lo_u32": "mul_lo", "v_mul_hi_u32": "mul_hi", "v_mad_u64_u32": "mad64", "v_mul_u32_u24": "mul24", "v_mad_u32_u24": "mul24",
    "v_lshlrev_b64": "shift64", "v_lshrrev_b64": "shift64", "v_lshl_add_u64": "lshl_add64", "v_alignbit_b32": "alignbit", "v_bfe_u32": "bfe", "v_and_or_b32": "and_or",
    "v_perm_b32": "perm", "v_bfrev_b32": "bfrev", "v_min_u32": "min32", "v_max_u32": "min32", "v_lshl_or_b32": "lshl_or", "v_xad_u32": "xad", "v_lshl_add_u32": "lshl_add32",
    "v_add_lshl_u32": "lshl_add32", "v_or3_b32": "or3", "v_ffbl_b32": "ffbl", "v_ffbh_u32": "ffbl", "v_mbcnt_lo_u32_b32": "mbcnt", "v_mbcnt_hi_u32_b32": "mbcnt",
    "v_readlane_b32": "lane", "v_writelane_b32": "lane", "v_readfirstlane_b32": "lane", "v_mov_b64": "mov64", "v_pk_mov_b32": "mov64",
    "v_add_co_u32": "addc64", "v_addc_co_u32": "addc64", "v_sub_co_u32": "addc64", "v_subb_co_u32": "addc64", "v_subrev_co_u32": "addc64", "v_subbrev_co_u32": "addc64",
    "v_cndmask_b32": "cmp32_cnd",
}
DEFAULT = "alignbit"          # anything unlisted: priced like the other three-operand / special integer instructions (4.2 - 4.4 cycles)


def rates(path):
    """key -> cycles per wave-instruction per SIMD at 4 waves per SIMD (256 CUs x 4 SIMDs x 2.4 GHz x 64 lanes / measured lane-ops/s)"""
    out = {}
    for l in open(path):
        m = re.search(r"key=(\S+)\s+8 waves/SIMD\s+([0-9.]+)\s+4 waves/SIMD\s+([0-9.]+)\s+1 wave/SIMD\s+([0-9.]+)", l)
        if m:
            out[m.group(1)] = 64.0 * 1024 * 2.4e9 / (float(m.group(3)) * 1e12)
    return out


def base(op):
    if op.startswith("v_cmp"):
        return "v_cmp64" if "64" in op else "v_cmp32"
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if op.endswith(suf):
            op = op[: -len(suf)]
    return op


def main():
    rate_file, out_file = sys.argv[1], sys.argv[2]
    # key[=pattern]: the JSON key bench.py looks up, and what the symbol must contain.  The instances that RUN are <.., BINS=false, DROP=false>: a bare
    # "scan_kernelILi31ELi17E" finds the scan-placed-items instance first (until the end of round 4 the scan was priced with THAT mix: 3.94 instead of 3.96 cycles)
    pats = sys.argv[3:] or ["scan_kernelILi31ELi17E=scan_kernelILi31ELi17ELb0ELb0E", "combine_kernelILi31E", "scan_kernelILi51ELi17E=scan_kernelILi51ELi17ELb0ELb0E"]
    cyc = rates(rate_file)
    asm = "/tmp/hsk_valu_floor.s"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    res = {"rates_file": os.path.relpath(rate_file, ROOT), "occupancy": "4 waves per SIMD column", "cycles_per_class": cyc, "kernels": {}}
    for spec in pats:
        key, _, pat = spec.partition("=")
        pat = pat or key
        st = [i for i, l in enumerate(lines) if re.match(r"^_ZN3hsk.*:", l) and pat in l]
        if not st:
            continue
        ins = []
        for l in lines[st[0] + 1:]:
            if l.startswith(".Lfunc_end"):
                break
            t = l.strip()
            if l.startswith("\t") and t and not t.startswith((".", ";")):
                ins.append(t.split()[0])
        valu = [i for i in ins if i.startswith("v_")]
        mix = Counter()
        for op in valu:
            b = base(op)
            if b == "v_cmp32":
                k = "cmp32"
            elif b == "v_cmp64":
                k = "cmp64_cnd"
            elif op.endswith("_dpp"):
                k = "dpp_add"
            else:
                k = CLASS.get(b, DEFAULT)
            mix[k] += 1
        tot = sum(mix.values())
        avg = sum(n * cyc.get(k, cyc[DEFAULT]) for k, n in mix.items()) / tot
        name = lines[st[0]].split(":")[0]
        res["kernels"][key] = {"symbol": name, "static_valu_instructions": tot, "static_salu_instructions": sum(1 for i in ins if i.startswith("s_")),
                               "static_lds_instructions": sum(1 for i in ins if i.startswith("ds_")), "mix": dict(mix.most_common()),
                               "avg_cycles_per_valu_wave_instruction": avg, "full_rate_share": sum(n for k, n in mix.items() if cyc.get(k, 9) < 3.2) / tot}
        print("%-28s %5d VALU instructions, %.2f cycles per wave-instruction on average (%.0f %% of them in the 2.3-2.8-cycle classes)" % (key, tot, avg, 100 * res["kernels"][key]["full_rate_share"]))
    json.dump(res, open(out_file, "w"), indent=1)


if __name__ == "__main__":
    main()
