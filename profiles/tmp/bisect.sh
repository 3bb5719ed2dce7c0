cp axtrack_amd/csrc/libaxtrack_hip.so /tmp/lib_asm.so
for v in GSTORE GLOAD; do
cp profiles/tmp/libaxt_$v.so axtrack_amd/csrc/libaxtrack_hip.so
echo "== variant $v"; NF=6 timeout -k 10 120 python profiles/tmp/dbg_conv0_ref.py 1024 1024 2>&1 | grep "^t="
done
cp /tmp/lib_asm.so axtrack_amd/csrc/libaxtrack_hip.so
