set -e
R=$GRAFT_REPO_ROOT
cd $R
run() { rm -rf gpurun_out/shards_repeat; bash tools/prof_shards_repeat.sh 2 "8:0 2:0" 2>&1 | grep -E "colsplit<8, 2, 4, 1, true" | awk -F'|' '{print "   ", $2, $6}'; }
echo "occupancy 4 (launch_bounds 256,4):"; run
sed -i 's/(LIGHT ? ((R == 8 \&\& U == 2 \&\& !PART) ? 4 : 2) : ((R == 8 \&\& U == 2) ? 4 : 1))/(LIGHT ? 2 : ((R == 8 \&\& U == 2) ? 4 : 1))/' conjugate-gradient_amd/csrc/cgx_kernels.hip
make -C conjugate-gradient_amd -s all > /dev/null 2>&1
echo "occupancy 3 (launch_bounds 256,2):"; run
echo "occupancy 4 again:"; git -C $R checkout conjugate-gradient_amd/csrc/cgx_kernels.hip 2>/dev/null || sed -i 's/(LIGHT ? 2 : ((R == 8 \&\& U == 2) ? 4 : 1))/(LIGHT ? ((R == 8 \&\& U == 2 \&\& !PART) ? 4 : 2) : ((R == 8 \&\& U == 2) ? 4 : 1))/' conjugate-gradient_amd/csrc/cgx_kernels.hip
make -C conjugate-gradient_amd -s all > /dev/null 2>&1; run
