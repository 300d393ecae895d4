# tools/valu_issue.sh -- runs tools/bin/valu_issue while sampling sclk; output -> gpurun_out/valu_issue.jsonl (+ clock samples)
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1 | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/valu_issue_clocks.txt &
SP=$!
python - <<'PY'
import torch, time
a = torch.randn(8192, 8192, device="cuda"); t0 = time.time()
while time.time() - t0 < 2.0:
    (a @ a); torch.cuda.synchronize()
PY
tools/bin/valu_issue 2400 > gpurun_out/valu_issue.jsonl
wait $SP
cat gpurun_out/valu_issue.jsonl
sort gpurun_out/valu_issue_clocks.txt | uniq -c
