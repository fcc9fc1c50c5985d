# usage: tools/ab_build.sh <name> <sed expression> [file]   -- builds ab/lib_<name>.so out of tree from the committed sources
# with one sed edit applied to csrc/<file> (default spconv_f32_ordered.hip); `base` with expression '' is the unmodified library.
# The working tree is not touched.  Run tools/ab_libs.sh on the GPU box afterwards.
name=$1; expr=$2; file=${3:-spconv_f32_ordered.hip}
root=$(cd $(dirname $0)/.. && pwd)
t=/tmp/abtree_$name
rm -rf $t && mkdir -p $t/sparsefusion_amd $t/include $root/ab || exit 1
cp -r $root/sparsefusion_amd/csrc $t/sparsefusion_amd/csrc && cp $root/include/*.h $t/include/ || exit 1
cd $t/sparsefusion_amd/csrc || exit 1
if [ -n "$expr" ]; then
  cp $file $file.orig
  sed -i "$expr" $file
  if cmp -s $file $file.orig; then echo "the edit changed nothing"; exit 1; fi
  rm $file.orig
fi
make -s -j8 LIB=$root/ab/lib_$name.so && ls -la $root/ab/lib_$name.so
