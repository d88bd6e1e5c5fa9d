# development aid: light-tracker frame rate (fuse_sequence 300 2) over workgroup width, group quantum and group count
# (libraries built by tools/icp_variants.sh build "t1024 -DVK_COLOR_THREADS=1024" ...)
for v in t1024 t512 t256; do
  mkdir -p /tmp/var_$v && cp vulcan_amd/lib/libvk_hip_var_$v.so /tmp/var_$v/libvk_hip.so
  for tg in ${TARGETS:-256}; do
    for q in ${QUANTA:-64 128 256}; do
      r=$(LD_LIBRARY_PATH=/tmp/var_$v:$LD_LIBRARY_PATH VK_COLOR_TARGET_GROUPS=$tg VK_COLOR_GROUP_QUANTUM=$q timeout -k 10 100 vulcan_amd/host/bin/fuse_sequence 300 2 | grep frames | sed 's/.*fps \([0-9.]*\).*/\1/')
      echo "$v target $tg quantum $q: $r fps"
    done
  done
done
