# build a variant of the library next to the default one: build_variant.sh <name> "<extra hipcc defs>"  -> xarray_parcel_amd/lib/lib_<name>.so
name=$1; shift
XP_EXTRA_DEFS="$*" XPARCEL_LIB=$PWD/xarray_parcel_amd/lib/lib_$name.so python -c "
from xarray_parcel_amd import _lib
_lib.build(force=True)
print('built', _lib.LIB_PATH)"
