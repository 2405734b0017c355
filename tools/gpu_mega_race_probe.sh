cd "$GRAFT_REPO_ROOT"
for p in 1 2 3; do N=150 B=6 python tools/gpu_mega_race_probe.py > /tmp/race_$p.txt 2>&1 & done; wait
for p in 1 2 3; do grep -v amdgpu.ids /tmp/race_$p.txt | head -30 | cut -c1-200; done
