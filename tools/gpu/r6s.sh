cd $GRAFT_REPO_ROOT
bash tools/gpu/call.sh r6s py tools/stress_parity.py 240 707 fused -- py tools/stress_parity.py 90 708
bash tools/gpu/profiles.sh r6s files
bash tools/gpu/profiles.sh r6s tuple
