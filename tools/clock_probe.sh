# tools/clock_probe.sh -- GPU clock / power while the extract+match bench runs (is the VALU-bound step power-throttled?)
python bench.py --steps 12000 --warmup 4 --no-ba --cpu-seconds 0 --no-pcie --no-exclusive > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
tail -1 /tmp/clk_bench.json | cut -c1-160
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo
