cd $GRAFT_REPO_ROOT
bash tools/gpu/evidence.sh r6ev2
bash tools/gpu/profiles.sh r6ev2 coeffs
bash tools/gpu/profiles.sh r6ev2 files
bash tools/gpu/profiles.sh r6ev2 tuple
