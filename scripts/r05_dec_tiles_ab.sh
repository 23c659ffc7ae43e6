cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { TSD_GEMM_CFG_OVERRIDE="$1" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg --no-sd15 --no-peaked --no-kloop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('override [$1] decode_ms', d['decode_ms'], 'encode_dev_ms', d['img2img_config4']['encode_ms_device'])"; }
for r in 1 2; do
run ""
run "2097152,128,1152:53"
run "2097152,128,1152:53;2097152,128,2304:53;2097152,128,1408:53"
run "2097152,128,1152:0"
run "524288,256,2304:2;524288,256,4608:2;524288,256,2816:2"
run "2097152,256,2304:2"
done
