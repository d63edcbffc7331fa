# register / scratch / occupancy report of the k_cape_cin instantiations (exact mode), also rebuilds the library
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage -o xarray_parcel_amd/lib/libxparcel.so xarray_parcel_amd/csrc/xparcel.hip 2>&1 | python3 -c "
import sys, re
name = None; rec = {}
for ln in sys.stdin:
    if 'error' in ln: print(ln.rstrip())
    m = re.search(r'Function Name: (\S+)', ln)
    if m: name = m.group(1); rec[name] = {}
    for key in ('VGPRs', 'AGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]'):
        m = re.search(r' ' + key + r': (\d+)', ln)
        if m and name: rec[name][key.split(' ')[0]] = int(m.group(1))
pat = sys.argv[1] if len(sys.argv) > 1 else 'k_cape_cinI[fd]Li[0-3]ELb[01]ELi0'
for n, r in rec.items():
    if re.search(pat, n): print(n[:60], r)
" "$@"
