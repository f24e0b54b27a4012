cp athenapk_amd/libapk_amd.so /tmp/new.so
for v in new prev; do
if [ $v = new ]; then cp /tmp/new.so athenapk_amd/libapk_amd.so; else cp athenapk_amd/libapk_amd_prev.so athenapk_amd/libapk_amd.so; fi
for mb in 16 32; do
OV="parthenon/mesh/nx1=128 parthenon/mesh/nx2=128 parthenon/mesh/nx3=128 parthenon/meshblock/nx1=$mb parthenon/meshblock/nx2=$mb parthenon/meshblock/nx3=$mb parthenon/time/nlim=20 parthenon/time/tlim=100"
echo "== $v: MHD PPM+HLLD VL2 128^3 in ${mb}^3 blocks"; python -m athenapk_amd -i synthetic_mhd -d /tmp/o $OV 2>&1 | tail -1
done; done
cp /tmp/new.so athenapk_amd/libapk_amd.so
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
