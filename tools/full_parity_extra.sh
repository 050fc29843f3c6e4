set -e
out=gpurun_out/r05h/full_parity_extra.txt
mkdir -p gpurun_out/r05h
echo "# tools/fuzz_full_parity.py <ring> <log2 degree> <batch>: EVERY word of the batch (adversarial patterns mixed in) against the oracle, final build of round 5 (source hash 00c35ef014778ea5)" > $out
for s in "goldilocks 20 256" "goldilocks 10 65536" "goldilocks 13 4096" "goldilocks 17 1024" "goldilocks 18 512" "goldilocks 19 256" "goldilocks 12 8192" "goldilocks 16 1024" "goldilocks 16 1096" "goldilocks 21 64" "babybear 20 128" "babybear 16 2056" "stark 16 64"; do
  echo "== $s" >> $out
  timeout -k 10 400 python3 tools/fuzz_full_parity.py $s 2>/dev/null | grep -v "^$" >> $out
done
tail -n 6 $out
