cd "${GRAFT_REPO_ROOT:-.}"
cp orb_slam3_rgbl_amd/librgbl_frontend.so /tmp/asbuilt.so
for v in 0 1 2; do
  cp exp/lib_skip$v.so orb_slam3_rgbl_amd/librgbl_frontend.so
  echo skip$v; bash tools/gpu_pmc.sh sk$v "SQ_INSTS_VALU SQ_INSTS_LDS" 2>&1 | grep k_fast
done
cp /tmp/asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so
