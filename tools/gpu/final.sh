cd $GRAFT_REPO_ROOT
bash tools/gpu/profiles.sh r6final coeffs
bash tools/gpu/profiles.sh r6final png
bash tools/gpu/profiles.sh r6final files
bash tools/gpu/profiles.sh r6final tuple
for f in gpurun_out/r6final/issue_*.json gpurun_out/r6final/traffic_*.json gpurun_out/r6final/issue_*_raw.csv gpurun_out/r6final/traffic_*_raw.csv; do cp $f profiles/; done
bash tools/gpu/evidence.sh r6final
