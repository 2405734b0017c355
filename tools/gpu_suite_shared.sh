#!/bin/bash
# The GPU test suite in two processes side by side on the one GPU: anything that only works with the device to itself shows up here.
cd "$GRAFT_REPO_ROOT"
K='not two_ranks and not rccl and not share_the_gpu and not splits'
python -m pytest tests -q -m gpu -k "$K" -p no:cacheprovider > /tmp/suite_a.txt 2>&1 &
python -m pytest tests -q -m gpu -k "$K" -p no:cacheprovider > /tmp/suite_b.txt 2>&1 &
wait
tail -4 /tmp/suite_a.txt | cut -c1-200; tail -4 /tmp/suite_b.txt | cut -c1-200
