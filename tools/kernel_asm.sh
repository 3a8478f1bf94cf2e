# device assembly + instruction mix of one kernel:  bash tools/kernel_asm.sh <file.hip> <substring of the mangled kernel name> [out.s]
R=$(cd "$(dirname "$0")/.." && pwd); SRC=$1; PAT=$2; OUT=${3:-/tmp/kernel.s}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -S --offload-device-only -o /tmp/_all.s $R/ci-gwas_amd/csrc/$SRC 2>&1 | grep -E "error|warning: [^a]"
python3 - "$PAT" "$OUT" <<'PY'
import re, collections, sys
s = open('/tmp/_all.s').read()
for k in [m for m in re.findall(r'\.amdhsa_kernel (\S+)', s) if sys.argv[1] in m]:
    i = s.index(k + ':'); j = s.index('.end_amdhsa_kernel', i); b = s[i:j]
    open(sys.argv[2], 'w').write(b)
    print(k[:100]); print(re.findall(r'\.amdhsa_next_free_vgpr \d+|\.amdhsa_next_free_sgpr \d+|\.amdhsa_private_segment_fixed_size \d+|\.amdhsa_group_segment_fixed_size \d+', b))
    ins = [l.split()[0] for l in b.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith('.') and not l.strip().startswith(';')]
    print(len(ins), collections.Counter(ins).most_common(22))
PY
