export HIP_FORCE_DEV_KERNARG=1; L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/keep.so; cp scripts/libtsd_new.so $L
for shape in "0,32,640,640,-1" "0,16,1280,1280,-1" "0,32,640,1280,-1" "1,32,640,640,-1" "1,16,1280,1280,-1"; do
  echo "== shape $shape epi=1"; TSD_GEMM_TS=1 TSD_BENCH_EPI=1 SHAPE=$shape python scripts/bench_gemm1.py 2>&1 | grep -E "\[ts\]|TF"
done; cp /tmp/keep.so $L
