# A/B of prebuilt library variants on the GPU box (build each variant here, copy libcusk_hip.so to variants/<name>.so --
# the directory is git-ignored but travels with gpurun -- and delete it afterwards):
#   gpurun -- 'bash tools/variants.sh variants/libA.so variants/libB.so'
for v in "$@"; do
  cp $v ci-gwas_amd/csrc/libcusk_hip.so
  echo "== $v"
  bash tools/l1_time.sh l1_exp=0 l1_exp=0 2>&1 | cut -c 1-100
done
