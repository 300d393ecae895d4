# the same soak stream with and without the device-side decision
cd $GRAFT_REPO_ROOT
python tools/soak_ba.py 40 2>&1 | grep -E "MISMATCH|soak_ba" | head -6
echo == no speculation
DVM_BA_NO_SPECULATION=1 python tools/soak_ba.py 40 2>&1 | grep -E "MISMATCH|soak_ba" | head -6
