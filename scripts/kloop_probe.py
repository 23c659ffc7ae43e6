"""K-loop slope / fixed cost of chosen tile configurations - the measurement behind bench.py's roofline.k_loop_model (four K lengths, least
squares, three repeats, spreads), run through the SAME function for any configuration list, so that the bench line and this probe agree by
construction when the box does.
   python scripts/kloop_probe.py "51,5,54,47" [conv|dense]      (no argument: bench.py's own probe list)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import bench, tsd
if len(sys.argv) > 1:
    cfgs = [int(c) for c in sys.argv[1].split(",")]
    kind = sys.argv[2] if len(sys.argv) > 2 else "conv"
    shapes = {"conv": [("conv C->320 @64x64 (M 32768)", 1, 8, 64, 64, 320, 320), ("conv C->640 @32x32 (M 8192)", 1, 8, 32, 32, 640, 640),
                       ("conv C->1280 @16x16 (M 2048)", 1, 8, 16, 16, 1280, 1280)],
              "dense": [("dense 8192x640xK", 0, 8, 32, 32, 640, 640), ("dense 2048x1280xK", 0, 8, 16, 16, 1280, 1280),
                        ("dense 32768x320xK", 0, 8, 64, 64, 320, 320)]}[kind]
    # (FM, FN, waves per SIMD, tile rows, tile columns are only used for the derived MFMA-clock figures: taken from bench.py's list where the
    # configuration is there, else left at the 128x160 4-wave tile)
    known = {p[7]: p[8:] for p in bench.K_LOOP_PROBES}
    probes = [(f"{lab}, cfg {c}",) + tuple(sh) + (c,) + tuple(known.get(c, (4, 5, 1, 128, 160))) for (lab, *sh) in shapes for c in cfgs]
else:
    probes = bench.K_LOOP_PROBES
for rec in bench.k_loop_model(tsd, 0, probes=probes):
    print(json.dumps(rec))
