#!/bin/bash
# tools/lab/pairs16/build.sh [extra hipcc flags, e.g. -DP16_KNOCK=4] -- from the repo root: a LAB copy of the library with the 16x16x32 pair kernel
# (gemm_pairs16.hip, reachable through grappa_gemm_desc.plan_cfg = 10 only) -> tools/lab/pairs16/libgrappa_hip_pairs16.so.  The shipped library and
# its sources are not touched: csrc is copied to tools/lab/pairs16/build, hooks.patch applied there.
set -e
L=tools/lab/pairs16
rm -rf $L/build && mkdir -p $L/build
cp grappa_amd/csrc/*.hip grappa_amd/csrc/*.h grappa_amd/csrc/Makefile $L/build/
cp $L/gemm_pairs16.hip $L/build/
(cd $L/build && patch -p1 -s < ../hooks.patch)
mkdir -p $L/include && cp include/*.h $L/include/            # (the sources say "../../include/...": csrc sits two levels below the root)
sed -i "s#^CXXFLAGS = #CXXFLAGS = $* #; s#\.\./\.\./include/#../include/#g; s#\.\./libgrappa_hip\.so#../libgrappa_hip_pairs16.so#g" $L/build/Makefile
sed -i 's#"\.\./\.\./include/#"../include/#' $L/build/*.h $L/build/*.hip
(cd $L/build && make -j8 ../libgrappa_hip_pairs16.so 2>&1 | grep -i " error\|Stop" ; true)
ls -la $L/libgrappa_hip_pairs16.so
