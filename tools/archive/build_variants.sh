#!/bin/bash
# usage: tools/build_variants.sh "<flags1>" "<flags2>" ...   -> tmp_variants/libphx_<i>.so + list.txt (+ register use of k_sssp_lds<2>)
cd /root/repo
rm -rf tmp_variants; mkdir -p tmp_variants
i=0
for v in "$@"; do
  i=$((i+1))
  make -s -C phanotate_amd/csrc clean
  make -s -C phanotate_amd/csrc EXTRA="$v -Rpass-analysis=kernel-resource-usage" 2> tmp_variants/res_$i.txt | grep -E "error"
  cp phanotate_amd/libphx.so tmp_variants/libphx_$i.so
  r=$(grep -A12 "Function Name: _Z10k_sssp_ldsILi2E" tmp_variants/res_$i.txt | grep -E "VGPRs:|Spill|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' ')
  echo "$i: $v  | $r" >> tmp_variants/list.txt
done
make -s -C phanotate_amd/csrc clean; make -s -C phanotate_amd/csrc
cat tmp_variants/list.txt
