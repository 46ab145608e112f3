for lib in "" ab/r4.so ab/prio.so "" ab/r4.so ab/prio.so; do
  if [ -n "$lib" ]; then export ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/$lib; else unset ALIGNNET_HIP_LIB; fi
  echo "lib=${lib:-tree}"; python tools/split_rate.py 1 | grep persistent
done
