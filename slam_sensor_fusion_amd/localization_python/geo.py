"""utm.from_latlon as the reference's Python twin calls it
(localization_python/localization_python/localization_node.py:138,
optimize_global_map_pose.py:43).  The `utm` package is an un-vendored, un-pinned dependency
of the reference and is not installed here; this restates its published series (utm 0.7.x:
Krueger/USGS coefficients with E = 0.00669438, hemisphere-aware false northing).  Parity
unpinned: no golden vector exists upstream; the oracle holds an independent C restatement."""
import math

K0 = 0.9996
E = 0.00669438
E2 = E * E
E3 = E2 * E
E_P2 = E / (1 - E)
M1 = 1 - E / 4 - 3 * E2 / 64 - 5 * E3 / 256
M2 = 3 * E / 8 + 3 * E2 / 32 + 45 * E3 / 1024
M3 = 15 * E2 / 256 + 45 * E3 / 1024
M4 = 35 * E3 / 3072
R = 6378137


def latlon_to_zone_number(latitude, longitude):
    if 56 <= latitude < 64 and 3 <= longitude < 12:
        return 32
    if 72 <= latitude <= 84 and longitude >= 0:
        if longitude < 9:
            return 31
        if longitude < 21:
            return 33
        if longitude < 33:
            return 35
        if longitude < 42:
            return 37
    return int((longitude + 180) / 6) % 60 + 1


def from_latlon(latitude, longitude):
    """-> (easting, northing, zone_number, zone_letter_is_north)"""
    lat_rad = math.radians(latitude)
    lat_sin, lat_cos = math.sin(lat_rad), math.cos(lat_rad)
    lat_tan = lat_sin / lat_cos
    lat_tan2 = lat_tan * lat_tan
    lat_tan4 = lat_tan2 * lat_tan2
    zone_number = latlon_to_zone_number(latitude, longitude)
    lon_rad = math.radians(longitude)
    central_lon_rad = math.radians((zone_number - 1) * 6 - 180 + 3)
    n = R / math.sqrt(1 - E * lat_sin ** 2)
    c = E_P2 * lat_cos ** 2
    a = lat_cos * ((lon_rad - central_lon_rad + math.pi) % (2 * math.pi) - math.pi)
    a2 = a * a
    a3 = a2 * a
    a4 = a3 * a
    a5 = a4 * a
    a6 = a5 * a
    m = R * (M1 * lat_rad - M2 * math.sin(2 * lat_rad) + M3 * math.sin(4 * lat_rad) - M4 * math.sin(6 * lat_rad))
    easting = K0 * n * (a + a3 / 6 * (1 - lat_tan2 + c) + a5 / 120 * (5 - 18 * lat_tan2 + lat_tan4 + 72 * c - 58 * E_P2)) + 500000
    northing = K0 * (m + n * lat_tan * (a2 / 2 + a4 / 24 * (5 - lat_tan2 + 9 * c + 4 * c ** 2)
                                        + a6 / 720 * (61 - 58 * lat_tan2 + lat_tan4 + 600 * c - 330 * E_P2)))
    if latitude < 0:
        northing += 10000000
    return easting, northing, zone_number, latitude >= 0
