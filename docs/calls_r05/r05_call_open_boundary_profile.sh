# round 5: kernel durations of the headline deck with boundary.field = Dirichlet and Open (the two open-boundary launches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_ob
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_ob -o kt -- python $R/scripts/open_boundary_rate.py 1024 96 once > $O/ob_rate.txt 2>$O/ob_err.txt
DB=$(find /tmp/prof_ob -name "*.db" | head -1)
python $R/scripts/kstats.py $DB 192 60 > $O/r05e_open_boundary_kstats.txt 2>>$O/ob_err.txt
cd $R
grep -i "multipole\|open_boundary\|rhs_all\|fillBuffer\|dst_rows" $O/r05e_open_boundary_kstats.txt | cut -c1-60,100-220
tail -3 $O/ob_err.txt
