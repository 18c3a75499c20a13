// Drives include/localization/localization_node.h like the reference's node is driven by its synchroniser
// (localization/src/localization_node.cpp:263-344): a map and a list of messages from files, one pose per line.
//   test_node_class map.bin messages.bin n_messages
// messages.bin, per message: int64 n_points, double compass_deg, gps (lat, lon, alt, cov[9]), odom (q_wxyz[4], t[3], cov[36]),
//                            float32 start pose[16] (row-major; applied before the message when its [15] == 1), float32 xyz[n_points * 3]
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "localization/localization_node.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    slamfusion::PointCloud map_cloud;
    {
        std::ifstream f(argv[1], std::ios::binary | std::ios::ate);
        const std::streamsize bytes = f.tellg();
        f.seekg(0);
        map_cloud.xyz.resize((std::size_t)bytes / sizeof(float));
        f.read(reinterpret_cast<char *>(map_cloud.xyz.data()), bytes);
    }
    slamfusion::Matrix4d mtg;
    {
        double rm[16];
        const double lla[3] = {-22.9068, -43.1729, 12.0};
        const float yaw = 0.0f;
        sf_fusion_map_T_global(lla, &yaw, 1, rm);
        mtg = slamfusion::Matrix4d::fromRowMajor(rm);
    }
    LocalizationCore core(map_cloud, mtg, {-22.9068, -43.1729, 12.0});
    core.setCoarseAlignmentComplete(true);
    std::ifstream f(argv[2], std::ios::binary);
    const int n_messages = atoi(argv[3]);
    for (int k = 0; k < n_messages; ++k) {
        int64_t n = 0;
        double compass = 0;
        sf_gps_fix gps;
        sf_odom odom;
        float start[16];
        f.read(reinterpret_cast<char *>(&n), sizeof(n));
        f.read(reinterpret_cast<char *>(&compass), sizeof(compass));
        f.read(reinterpret_cast<char *>(&gps), sizeof(double) * 12);
        f.read(reinterpret_cast<char *>(&odom), sizeof(double) * 43);
        f.read(reinterpret_cast<char *>(start), sizeof(start));
        slamfusion::PointCloud scan;
        scan.xyz.resize((std::size_t)n * 3);
        f.read(reinterpret_cast<char *>(scan.xyz.data()), (std::streamsize)(sizeof(float) * scan.xyz.size()));
        if (start[15] == 1.0f) core.setPose(slamfusion::Matrix4f::fromRowMajor(start));
        core.compassCallback(compass);
        slamfusion::Matrix4f pose;
        const bool ok = core.localizationCallback(scan, gps, odom, pose);
        std::printf("%d %d %lld", (int)ok, core.last().icp.iterations, (long long)core.last().n_scan);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) std::printf(" %.9g", (double)pose(i, j));
        std::printf("\n");
    }
    return 0;
}
