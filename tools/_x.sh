for X in timing; do echo "== $X"; MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_$X.so python tests/vp_resident_timeline.py 128 2>&1 | tail -6 | cut -c1-400; done
