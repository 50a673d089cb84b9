# usage (GPU box): bash tools/h2h_ab.sh <vA> <vB>  -> tools/h2h_parts.py with tmp_variants/libphx_<v>.so, alternating
cd /root/repo
cp phanotate_amd/libphx.so /tmp/d.so
for rep in 1 2 3; do for v in $1 $2; do cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so; echo -n "$v "; python tools/h2h_parts.py 2>/dev/null | head -1; done; done
cp /tmp/d.so phanotate_amd/libphx.so
