# builds tools/probe/libhmvit_base.so from the sources of a git revision (default HEAD) for same-box A/B runs (ab.sh)
set -e
REV=${1:-HEAD}
D=/tmp/hmvit_base_src
rm -rf $D && mkdir -p $D/hm-vit_amd/csrc $D/include
for f in $(git ls-tree --name-only $REV hm-vit_amd/csrc/ | grep -E "\.(hip|hpp)$|Makefile$"); do git show $REV:$f > $D/$f; done
git show $REV:include/hmvit.h > $D/include/hmvit.h
make -C $D/hm-vit_amd/csrc -j4 > /dev/null
cp $D/hm-vit_amd/libhmvit.so tools/probe/libhmvit_base.so
echo "tools/probe/libhmvit_base.so <- $REV"
