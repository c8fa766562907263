#!/bin/bash
# sample the shader clock / power while the gate-up GEMM runs back to back (is the matrix pipe clock-throttled under this load?)
cd "$(dirname "$0")"
(timeout 60 ./bench_gemm 10922 400 > /dev/null 2>&1) &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo
  sleep 0.7
done
kill $BP 2>/dev/null
wait $BP 2>/dev/null
echo idle:
sleep 1
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
