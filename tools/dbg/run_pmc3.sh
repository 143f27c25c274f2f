mkdir -p gpurun_out
for sh in "fwd 65536 512 128" "fwd 23894 128 512" "fwd 1450 2048 512" "dgrad 65536 512 128"; do
  tag=$(echo $sh | tr ' ' '_')
  timeout 400 bash tools/dbg/pmc_one.sh $tag $sh fp32 > gpurun_out/pmc3_$tag.txt 2>&1
  python tools/dbg/one_gemm.py $sh fp32 2>/dev/null | tail -1 >> gpurun_out/pmc3_$tag.txt
  rm -rf gpurun_out/pmc1_${tag}_*
  cat gpurun_out/pmc3_$tag.txt
done
