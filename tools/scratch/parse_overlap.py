"""Experiment: does place_kernel overlap with scan_kernel when the scan leaves one workgroup slot per CU free?  (hsk_debug_parse_overlap)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
G = int(312_500_000 * scale); NR = G * 32 // 150
c = H.Context(K=31, M=17, L=15, U=40, ntasks=40)
dp, nb, do, dl = c.synth_reads(G, 150, NR, 20251003)
c.lib.hsk_debug_parse_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
for blocks in (1024, 768, 512):
    out = (C.c_double * 6)()
    rc = c.lib.hsk_debug_parse_overlap(c.h, dp, nb, do, dl, NR, blocks, out)
    print("blocks %4d rc %d: scan(1024) %.2f  scan(blocks) %.2f  place alone %.2f  both at once %.2f (scan %.2f, place %.2f)  -> serial %.2f" % (
        blocks, rc, out[0], out[1], out[2], out[3], out[4], out[5], out[1] + out[2]))
