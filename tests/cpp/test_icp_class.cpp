// Drives the reference-shaped C++ class (include/localization/*.h) exactly like
// LocalizationNode does (localization/src/localization_node.cpp:24-29,292-296,335-338):
// reads a map and a scan (raw float32 xyz files), prints the ICPResult as one line.
//   test_icp_class map.bin scan.bin max_dist iters accept eps
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "localization/icp_point_to_point.h"
#include "localization/point_cloud_processing.hpp"
#include "localization/stochastic_filter.h"

static slamfusion::PointCloud::Ptr read_cloud(const char *path)
{
    auto c = std::make_shared<slamfusion::PointCloud>();
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    c->xyz.resize((std::size_t)bytes / sizeof(float));
    f.read(reinterpret_cast<char *>(c->xyz.data()), bytes);
    return c;
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    auto map_cloud = read_cloud(argv[1]);
    auto scan_cloud = read_cloud(argv[2]);
    applyUniformSubsample(scan_cloud, 2);                                          // localization_node.cpp:292
    auto cropped = std::make_shared<slamfusion::PointCloud>();
    cropPointCloudThroughRadius(slamfusion::Matrix4f::Identity(), 10.0, scan_cloud, cropped);   // :296
    auto icp = std::make_shared<ICPPointToPoint>((float)atof(argv[3]), atoi(argv[4]), (float)atof(argv[5]), (float)atof(argv[6]));   // :24-28
    icp->setDebugMode(false);
    icp->setTargetPointCloud(map_cloud);                                           // :303
    icp->setSourcePointCloud(cropped);                                             // :335
    icp->setInitialTransformation(slamfusion::Matrix4f::Identity());               // :336
    const auto r = icp->calculateAlignment();                                      // :337
    std::printf("%zu %d %d %.9g", cropped->size(), r.iterations, (int)r.has_converged, (double)r.error);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) std::printf(" %.9g", (double)r.transformation(i, j));
    std::printf("\n");
    StochasticFilter filter(4, 3.0f);
    filter.addPoseToQueue(r.transformation);
    (void)filter.applyGaussianFilterToCurrentPose(slamfusion::Matrix4f::Identity(), r.transformation);
    return 0;
}
