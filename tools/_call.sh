set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s
rm -rf $O; mkdir -p $O
for T in t1 t2 full; do
  if [ $T = full ]; then L=$PWD/mvsmplfitting_amd/libmvfit.so; else L=$PWD/mvsmplfitting_amd/libmvfit_$T.so; fi
  echo $T; MVFIT_LIBRARY=$L PYTHONPATH=. timeout 300 python tests/report_vertex_pass.py > $O/vp_$T.log 2>&1; grep "^B " $O/vp_$T.log
done
