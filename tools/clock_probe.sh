# tools/clock_probe.sh -- GPU clock / power while the extract+match bench runs (is the VALU-bound step power-throttled?)
python bench.py --steps 150 --warmup 4 --no-legs --no-ba --cpu-seconds 0 --no-pcie --no-exclusive > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 3
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*(\([0-9]*Mhz\)).*/\1/' -e 's/.*(W): //' | tr '\n' ' '; echo
  kill -0 $BP 2>/dev/null || break
  sleep 0.3
done
wait $BP
tail -1 /tmp/clk_bench.json | cut -c1-160
