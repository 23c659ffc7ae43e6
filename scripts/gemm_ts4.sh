cp scripts/libtsd_ts.so stable-diffusion.mojo_amd/lib/libtsd.so
for shape in "1,64,320,320,0" "1,64,320,320,30" "1,64,640,320,0" "1,128,512,512,2" "1,128,512,512,32"; do
  echo "== shape $shape"; TSD_GEMM_TS=1 SHAPE=$shape python scripts/bench_gemm1.py 2>&1 | grep -E "main loop|in the main|TF"
done
