#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do for v in 2 1 4 8; do
  echo -n "TSD_GN_APPLY_MULT=$v: "
  TSD_GN_APPLY_MULT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg --no-img2img --no-sd15 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['decode_ms'], d['vae_roofline']['decoder'])"
done; done
