#!/bin/bash
# samples the shader clock / power while bench.py is running (is the MFMA peak of 2.4 GHz actually available under this load?)
mkdir -p gpurun_out
python bench.py --steps 1500 --warmup 5 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy" 
  echo ---
  sleep 0.7
done > gpurun_out/clk_samples.log 2>&1
wait $BP
tail -2 gpurun_out/clk_bench.log | cut -c1-600
cat gpurun_out/clk_samples.log | head -60
