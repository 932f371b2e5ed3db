cd /root/repo
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for r in 1 2; do
for v in k2w16 k2w12; do
  cp tools/_ab/libmbavo_$v.so mba-vo_amd/libmbavo.so
  for fp in 1 0; do
    echo "== $v FUSED_POSE=$fp"; MBAVO_FUSED_POSE=$fp python tools/mode_bench.py 2>&1 | grep "k=2 dense"
  done
done
done
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
