// ref_geo_shim.cpp — TEST INFRASTRUCTURE (oracle).
// Builds oracle/_ref/libsfref.so from the ONE reference file that compiles stand-alone:
// /root/reference/localization/include/localization/geo_lib.hpp (libm only).  The header
// is included from where it lies (never copied into this repo); this shim only gives
// UTM::LLtoUTM a C symbol so tests can pin oracle/fusion.c:orc_ll_to_utm against it.
#include <localization/geo_lib.hpp>

extern "C" void sfref_ll_to_utm(double lat, double lon, double *northing, double *easting)
{
    UTM::LLtoUTM(lat, lon, *northing, *easting);
}
