cp scripts/libtsd_ts.so stable-diffusion.mojo_amd/lib/libtsd.so
for shape in "1,64,320,320,0" "1,64,320,320,5" "1,64,320,320,14" "1,64,320,320,11" "1,64,640,320,0" "1,64,640,320,14"; do
  echo "== shape $shape"; TSD_GEMM_TS=1 SHAPE=$shape python scripts/bench_gemm1.py 2>&1 | grep -E "\[ts\]|TF"
done
