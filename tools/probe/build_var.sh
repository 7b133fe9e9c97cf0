# builds tools/probe/lib_<name>.so from the CURRENT sources with extra compiler flags (same-box A/B runs through HMVIT_LIB)
#   bash tools/probe/build_var.sh nomath "-DHMVIT_EXP_PCS_NOMATH" [file.hip ...]   (default: all sources)
set -e
NAME=$1; FLAGS=$2; shift 2 || true
D=/tmp/hmvit_var_$NAME
rm -rf $D && mkdir -p $D/hm-vit_amd/csrc $D/include
cp hm-vit_amd/csrc/*.hip hm-vit_amd/csrc/*.hpp hm-vit_amd/csrc/Makefile $D/hm-vit_amd/csrc/
cp include/hmvit.h $D/include/
# objects that do not depend on the flags are reused from the tree
if [ $# -gt 0 ]; then
  cp hm-vit_amd/csrc/*.o $D/hm-vit_amd/csrc/ 2>/dev/null || true
  for f in "$@"; do rm -f $D/hm-vit_amd/csrc/${f%.hip}.o; done
  touch -d "2000-01-01" $D/hm-vit_amd/csrc/*.hpp $D/include/hmvit.h $D/hm-vit_amd/csrc/*.hip
  for f in "$@"; do touch $D/hm-vit_amd/csrc/$f; done
  touch $D/hm-vit_amd/csrc/*.o; for f in "$@"; do rm -f $D/hm-vit_amd/csrc/${f%.hip}.o; done
fi
make -C $D/hm-vit_amd/csrc -j4 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable -DHMVIT_ALLOW_EXP $FLAGS" > /dev/null
cp $D/hm-vit_amd/libhmvit.so tools/probe/lib_$NAME.so
echo "tools/probe/lib_$NAME.so <- current sources + $FLAGS"
