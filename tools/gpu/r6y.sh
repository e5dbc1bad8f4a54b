cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python tools/mt_first_calls.py 3 2>&1 | tail -1; done
for i in 1 2 3 4 5 6; do PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/mt_first_calls.py 3 2>&1 | tail -1; done
for i in 1 2 3; do python tools/mt_first_calls.py 1 2>&1 | tail -1; done
for i in 1 2 3; do python tools/mt_first_calls.py 3 gradient 2>&1 | tail -1; done
