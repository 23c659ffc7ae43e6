#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/keep.so
for rep in 1 2 3; do
echo "== new, fused on ($rep)"; timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "img2img_config4 or headline_size_properties" 2>&1 | grep -E "Mismatch|Max abs|passed|failed" | head -5
done
echo "== new, TSD_CHAIN=0"; TSD_CHAIN=0 timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "img2img_config4 or headline_size_properties" 2>&1 | grep -E "Mismatch|Max abs|passed|failed" | head -5
echo "== new, TSD_CHAIN=0 (2)"; TSD_CHAIN=0 timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "img2img_config4 or headline_size_properties" 2>&1 | grep -E "Mismatch|Max abs|passed|failed" | head -5
cp scripts/libtsd_base.so $L
for rep in 1 2 3; do
echo "== round-3 build ($rep)"; timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "img2img_config4 or headline_size_properties" 2>&1 | grep -E "Mismatch|Max abs|passed|failed" | head -5
done
cp /tmp/keep.so $L
