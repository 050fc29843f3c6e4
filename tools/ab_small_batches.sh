# A/B of two library builds on small batches: default plan only
mkdir -p gpurun_out/r04p
for r in 1 2; do
  for lib in "$PWD/build_tmp/libsr_before_onestream_pair.so" ""; do
    if [ -n "$lib" ]; then export SR_LIB_PATH=$lib; tag=other; else unset SR_LIB_PATH; tag=tree; fi
    echo "== $tag" >> gpurun_out/r04p/ab_small.txt
    python3 tools/bench_small_batches.py goldilocks 16 32 goldilocks 16 64 goldilocks 16 128 goldilocks 16 256 goldilocks 20 4 goldilocks 20 16 goldilocks 18 32 2>/dev/null | grep "^| [gb]" | cut -d'|' -f2-6 >> gpurun_out/r04p/ab_small.txt
  done
done
cat gpurun_out/r04p/ab_small.txt
