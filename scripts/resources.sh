# register / scratch / occupancy report of the k_cape_cin instantiations of one translation unit
# usage: resources.sh [T=double] [MODE=0] [name regex]
T=${1:-double}; MODE=${2:-0}; PAT=${3:-k_cape_cin}
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -DXP_TU_T=$T -DXP_TU_MODE=$MODE -Rpass-analysis=kernel-resource-usage \
  -o /tmp/xp_cape_tu_res.o xarray_parcel_amd/csrc/xp_cape_tu.hip 2>&1 | python3 -c "
import sys, re
name = None; rec = {}
for ln in sys.stdin:
    if 'error' in ln: print(ln.rstrip())
    m = re.search(r'Function Name: (\S+)', ln)
    if m: name = m.group(1); rec[name] = {}
    for key in ('VGPRs', 'AGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]'):
        m = re.search(r' ' + key + r': (\d+)', ln)
        if m and name: rec[name][key.split(' ')[0]] = int(m.group(1))
for n, r in rec.items():
    if re.search(sys.argv[1], n): print(n[:64], r)
" "$PAT"
