"""GPU evidence that licenses `isa_lint.py --update`: on exactly the in-tree library, RACE_LOOPS (>= 1000) identical
50-step denoise loops at the headline size must give ONE result, and JITTER_LOOPS (>= 300) loops of the -DTSD_JITTER build (random
wave-level delays at every tile step, barrier and hand-off) must reproduce those bits.  Writes gpurun_out/bless_record.json with the
sha256 of the library's gfx950 code objects; copy it to profiles/bless_record.json and run the lint's --update."""
import ast, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd", "tools"))
import isa_lint
LIB = os.path.join(ROOT, "stable-diffusion.mojo_amd", "lib", "libtsd.so")
JIT = os.path.join(ROOT, "stable-diffusion.mojo_amd", "lib", "libtsd_jitter.so")
RACE = int(os.environ.get("RACE_LOOPS", 1000)); JITTER = int(os.environ.get("JITTER_LOOPS", 300))


def loops(lib, n):
    env = dict(os.environ, TSD_LIB=lib, N=str(n))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "diag_race5.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=3000).stdout
    m = re.search(r"(\d+) distinct (\{.*\})", out)
    assert m, out[-2000:]
    return ast.literal_eval(m.group(2)), out.strip().splitlines()[-1]


t0 = time.time()
ref, line_ref = loops(LIB, RACE)
jit, line_jit = loops(JIT, JITTER)
rec = {"code_sha256": isa_lint.code_sha256(LIB), "jitter_code_sha256": isa_lint.code_sha256(JIT),
       "race_loops": sum(ref.values()), "race_distinct": len(ref), "jitter_loops": sum(jit.values()),
       "jitter_matches_shipped": len(jit) == 1 and set(jit) == set(ref), "result": sorted(ref), "jitter_result": sorted(jit),
       "shipped_run": line_ref, "jitter_run": line_jit, "hipcc": isa_lint.hipcc_version(), "seconds": round(time.time() - t0, 1),
       "date": time.strftime("%Y-%m-%d %H:%M:%S")}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "bless_record.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
sys.exit(0 if rec["race_distinct"] == 1 and rec["jitter_matches_shipped"] else 1)
