# timing probes of rns_extend_mfma_kernel: tools/prof_mfma_ext.py against the default library, the VALU form and the
# variant libraries named on the command line (tools/mfma_variant.sh)
for v in mfma valu "$@"; do
  unset HX_NO_MFMA_EXT HX_LIB HX_HOST_LIB
  case $v in
    valu) export HX_NO_MFMA_EXT=1;;
    mfma) ;;
    *) export HX_LIB=$PWD/helib_amd/lib/variants/$v/libhelib_amd.so HX_HOST_LIB=$PWD/helib_amd/lib/variants/$v/libhelib_amd_host.so;;
  esac
  echo "== $v"; timeout 300 python tools/prof_mfma_ext.py 2>&1 | tail -3
done
