// sf_io.cpp — on-disk formats either side of the registration path and the
// GlobalMapFramesManager start-up logic, host C++.
//
// Restates localization/src/global_map_frames_manager.cpp:8-67 (text parsers), :69-91
// (altitude table), :93-151 (map.pcd cache / merge + voxel grid + save), :153-248 (reading
// filter, map_T_global) and the files mapping/src/map_data_save_node.cpp:23-29,70-98 writes
// (PCD v0.7 "DATA binary" tiles via pcl::io::savePCDFileBinary, "tx ty tz" / "lat lon alt y"
// text logs).  The voxel grid of the merge runs on the device (sf_cloud_voxel_downsample).
#include "sf_common.hpp"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <dirent.h>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };
constexpr int64_t PCD_MAX_POINTS = (int64_t)1 << 31;          // far beyond any tile the recorder writes
constexpr int64_t PCD_MAX_BYTES = (int64_t)64 << 30;          // raw payload cap (host memory)

// nothing may unwind through the extern "C" boundary (a ctypes / ROS host would std::terminate)
template <class F> int guarded(F &&fn)
{
    try {
        return fn();
    } catch (const std::bad_alloc &) {
        sf::set_error("out of host memory");
        return SF_ERR_NOMEM;
    } catch (const std::exception &e) {
        sf::set_error("I/O failure: %s", e.what());
        return SF_ERR_INVALID;
    } catch (...) {
        sf::set_error("I/O failure");
        return SF_ERR_INVALID;
    }
}

double read_scalar(const unsigned char *p, const PcdField &f)
{
    switch (f.type) {
    case 'F':
        if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
        if (f.size == 8) { double v; std::memcpy(&v, p, 8); return v; }
        break;
    case 'I':
        if (f.size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; }
        if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
        if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
        if (f.size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
        break;
    case 'U':
        if (f.size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; }
        if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
        if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
        if (f.size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
        break;
    }
    return NAN;
}

// LZF decompression (PCD "binary_compressed"); returns bytes written or 0 on error
size_t lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    const unsigned char *ip = in, *in_end = in + in_len;
    unsigned char *op = out, *out_end = out + out_len;
    while (ip < in_end) {
        unsigned ctrl = *ip++;
        if (ctrl < 32) {
            ++ctrl;
            if (op + ctrl > out_end || ip + ctrl > in_end) return 0;
            std::memcpy(op, ip, ctrl);
            op += ctrl;
            ip += ctrl;
        } else {
            unsigned len = ctrl >> 5;
            if (len == 7) { if (ip >= in_end) return 0; len += *ip++; }
            if (ip >= in_end) return 0;
            const unsigned char *ref = op - ((ctrl & 0x1f) << 8) - 1 - *ip++;
            if (ref < out || op + len + 2 > out_end) return 0;
            for (unsigned k = 0; k < len + 2; ++k) *op++ = *ref++;
        }
    }
    return (size_t)(op - out);
}

int pcd_read(const std::string &path, std::vector<float> &xyz)
{
    std::ifstream f(path, std::ios::binary);
    SF_CHECK(f.is_open(), SF_ERR_INVALID, "cannot open %s", path.c_str());
    std::vector<PcdField> fields;
    int64_t width = 0, height = 1, points = -1;
    bool have_points = false;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        std::string key;
        is >> key;
        if (key == "FIELDS") { std::string n; while (is >> n) { PcdField pf; pf.name = n; fields.push_back(pf); } }
        else if (key == "SIZE") { for (auto &pf : fields) is >> pf.size; }
        else if (key == "TYPE") { for (auto &pf : fields) is >> pf.type; }
        else if (key == "COUNT") { for (auto &pf : fields) is >> pf.count; }
        else if (key == "WIDTH") is >> width;
        else if (key == "HEIGHT") is >> height;
        else if (key == "POINTS") { is >> points; have_points = !is.fail(); }
        else if (key == "DATA") { is >> data_kind; break; }
    }
    SF_CHECK(!fields.empty() && !data_kind.empty(), SF_ERR_INVALID, "%s: not a PCD file", path.c_str());
    // the header is untrusted input: every number that sizes a buffer or an offset is checked
    SF_CHECK(width >= 0 && height >= 0 && (height == 0 || width <= PCD_MAX_POINTS / height), SF_ERR_INVALID, "%s: bad WIDTH / HEIGHT", path.c_str());
    if (!have_points) points = width * height;
    SF_CHECK(points >= 0 && points <= PCD_MAX_POINTS, SF_ERR_INVALID, "%s: bad POINTS %lld", path.c_str(), (long long)points);
    int64_t step = 0;
    int ix = -1, iy = -1, iz = -1;
    for (size_t k = 0; k < fields.size(); ++k) {
        const PcdField &pf = fields[k];
        SF_CHECK((pf.size == 1 || pf.size == 2 || pf.size == 4 || pf.size == 8) && pf.count >= 1 && pf.count <= 4096 && (pf.type == 'F' || pf.type == 'I' || pf.type == 'U'),
                 SF_ERR_INVALID, "%s: bad SIZE / TYPE / COUNT of field %s", path.c_str(), pf.name.c_str());
        fields[k].offset = (int)step;
        step += (int64_t)pf.size * pf.count;
        SF_CHECK(step <= (1 << 20), SF_ERR_INVALID, "%s: point step too large", path.c_str());
        if (pf.name == "x") ix = (int)k;
        if (pf.name == "y") iy = (int)k;
        if (pf.name == "z") iz = (int)k;
    }
    SF_CHECK(ix >= 0 && iy >= 0 && iz >= 0, SF_ERR_INVALID, "%s: no x/y/z fields", path.c_str());
    for (int k : {ix, iy, iz})
        SF_CHECK(fields[(size_t)k].type == 'F' && (fields[(size_t)k].size == 4 || fields[(size_t)k].size == 8), SF_ERR_INVALID, "%s: x/y/z must be F4 or F8", path.c_str());
    SF_CHECK(points == 0 || step <= PCD_MAX_BYTES / points, SF_ERR_INVALID, "%s: %lld points x %lld bytes is too large", path.c_str(), (long long)points, (long long)step);
    xyz.resize((size_t)points * 3);
    if (data_kind == "ascii") {
        for (int64_t i = 0; i < points; ++i) {
            SF_CHECK((bool)std::getline(f, line), SF_ERR_INVALID, "%s: truncated ascii data", path.c_str());
            std::istringstream is(line);
            int col = 0;
            for (size_t k = 0; k < fields.size(); ++k)
                for (int c = 0; c < fields[k].count; ++c, ++col) {
                    std::string tok;
                    is >> tok;
                    const float v = (tok == "nan" || tok == "NaN") ? NAN : (float)std::strtod(tok.c_str(), nullptr);
                    if ((int)k == ix && c == 0) xyz[3 * (size_t)i] = v;
                    if ((int)k == iy && c == 0) xyz[3 * (size_t)i + 1] = v;
                    if ((int)k == iz && c == 0) xyz[3 * (size_t)i + 2] = v;
                }
        }
        return SF_OK;
    }
    std::vector<unsigned char> raw;
    if (data_kind == "binary") {
        raw.resize((size_t)points * (size_t)step);
        f.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size());
        SF_CHECK((size_t)f.gcount() == raw.size(), SF_ERR_INVALID, "%s: truncated binary data", path.c_str());
        for (int64_t i = 0; i < points; ++i) {
            const unsigned char *p = raw.data() + (size_t)i * (size_t)step;
            xyz[3 * (size_t)i] = (float)read_scalar(p + fields[ix].offset, fields[ix]);
            xyz[3 * (size_t)i + 1] = (float)read_scalar(p + fields[iy].offset, fields[iy]);
            xyz[3 * (size_t)i + 2] = (float)read_scalar(p + fields[iz].offset, fields[iz]);
        }
        return SF_OK;
    }
    if (data_kind == "binary_compressed") { // LZF block, fields stored one after the other (SoA)
        uint32_t comp = 0, uncomp = 0;
        f.read(reinterpret_cast<char *>(&comp), 4);
        SF_CHECK(f.gcount() == 4, SF_ERR_INVALID, "%s: truncated compressed header", path.c_str());
        f.read(reinterpret_cast<char *>(&uncomp), 4);
        SF_CHECK(f.gcount() == 4, SF_ERR_INVALID, "%s: truncated compressed header", path.c_str());
        SF_CHECK((size_t)uncomp == (size_t)points * (size_t)step, SF_ERR_INVALID, "%s: compressed size mismatch", path.c_str());
        // LZF cannot expand: a stream longer than its output (+ control bytes) is not a PCD block
        SF_CHECK((uint64_t)comp <= (uint64_t)uncomp + (uint64_t)uncomp / 16 + 64, SF_ERR_INVALID, "%s: compressed size %u is implausible", path.c_str(), comp);
        std::vector<unsigned char> cbuf(comp);
        f.read(reinterpret_cast<char *>(cbuf.data()), comp);
        SF_CHECK((uint32_t)f.gcount() == comp, SF_ERR_INVALID, "%s: truncated compressed data", path.c_str());
        raw.resize(uncomp);
        SF_CHECK(lzf_decompress(cbuf.data(), comp, raw.data(), uncomp) == uncomp, SF_ERR_INVALID, "%s: LZF stream corrupt", path.c_str());
        size_t base = 0;
        std::vector<size_t> field_base(fields.size());
        for (size_t k = 0; k < fields.size(); ++k) { field_base[k] = base; base += (size_t)fields[k].size * fields[k].count * (size_t)points; }
        for (int64_t i = 0; i < points; ++i) {
            xyz[3 * (size_t)i] = (float)read_scalar(raw.data() + field_base[ix] + (size_t)i * fields[ix].size * fields[ix].count, fields[ix]);
            xyz[3 * (size_t)i + 1] = (float)read_scalar(raw.data() + field_base[iy] + (size_t)i * fields[iy].size * fields[iy].count, fields[iy]);
            xyz[3 * (size_t)i + 2] = (float)read_scalar(raw.data() + field_base[iz] + (size_t)i * fields[iz].size * fields[iz].count, fields[iz]);
        }
        return SF_OK;
    }
    sf::set_error("%s: unsupported DATA kind '%s'", path.c_str(), data_kind.c_str());
    return SF_ERR_INVALID;
}

// what pcl::io::savePCDFileBinary writes for pcl::PointXYZ (PCD v0.7, packed x y z)
int pcd_write_binary(const std::string &path, const float *xyz, int64_t n)
{
    std::ofstream f(path, std::ios::binary);
    SF_CHECK(f.is_open(), SF_ERR_INVALID, "cannot create %s", path.c_str());
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH " << n
      << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    f.write(reinterpret_cast<const char *>(xyz), (std::streamsize)(sizeof(float) * 3 * (size_t)n));
    SF_CHECK(f.good(), SF_ERR_INVALID, "write to %s failed", path.c_str());
    return SF_OK;
}

} // namespace

extern "C" int sf_pcd_read(const char *path, float **xyz, int64_t *n)
{
    SF_CHECK(path && xyz && n, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        std::vector<float> v;
        SF_TRY(pcd_read(path, v));
        *n = (int64_t)(v.size() / 3);
        *xyz = (float *)std::malloc(sizeof(float) * std::max<size_t>(v.size(), 1));
        SF_CHECK(*xyz, SF_ERR_NOMEM, "out of host memory");
        std::memcpy(*xyz, v.data(), sizeof(float) * v.size());
        return SF_OK;
    });
}

extern "C" void sf_free(void *p) { std::free(p); }

extern "C" int sf_pcd_write_binary(const char *path, const float *xyz, int64_t n)
{
    SF_CHECK(path && n >= 0 && (xyz || n == 0), SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int { return pcd_write_binary(path, xyz, n); });
}

extern "C" int sf_cloud_load_pcd(sf_cloud *c, const char *path)
{
    SF_CHECK(c && path, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        std::vector<float> v;
        SF_TRY(pcd_read(path, v));
        return sf_cloud_upload(c, v.data(), (int64_t)(v.size() / 3));
    });
}

extern "C" int sf_cloud_save_pcd(sf_cloud *c, const char *path)
{
    SF_CHECK(c && path, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        std::vector<float> v((size_t)c->n * 3);
        SF_TRY(sf_cloud_download(c, v.data(), c->n, nullptr));
        return pcd_write_binary(path, v.data(), c->n);
    });
}

// ------------------------------------------------------------------ GlobalMapFramesManager
struct sf_frames {
    std::string data_folder, map_name;
    std::size_t num_poses_max = 0;
    std::vector<double> altitude_table; // lat, lon, alt rows with alt > 0 (:60-63)
};

namespace {

// global_map_frames_manager.cpp:8-34
std::vector<double> load_odometry_positions(const std::string &file)
{
    std::vector<double> pos;
    std::ifstream f(file);
    if (!f.is_open()) return pos;
    std::string line;
    while (std::getline(f, line)) {
        if (line == "tx ty tz") continue;
        std::istringstream is(line);
        double x = 0, y = 0, z = 0;
        is >> x >> y >> z;
        pos.push_back(x); pos.push_back(y); pos.push_back(z);
    }
    return pos;
}

// global_map_frames_manager.cpp:36-67; yaw parsed as float like the reference
void load_global_info(const std::string &file, std::vector<double> &lla, std::vector<float> &yaw, std::vector<double> &table)
{
    std::ifstream f(file);
    if (!f.is_open()) return;
    std::string line;
    while (std::getline(f, line)) {
        if (line == "lat lon alt y") continue;
        std::istringstream is(line);
        double la = 0, lo = 0, al = 0;
        float y = 0;
        is >> la >> lo >> al >> y;
        lla.push_back(la); lla.push_back(lo); lla.push_back(al);
        yaw.push_back(y);
        if (al > 0) { table.push_back(la); table.push_back(lo); table.push_back(al); }
    }
}

bool ends_with_pcd(const std::string &n) { return n.size() > 4 && n.substr(n.size() - 4) == ".pcd"; }

} // namespace

extern "C" sf_frames *sf_frames_create(const char *data_folder, const char *map_name, int64_t num_poses_max)
{
    if (!data_folder || !map_name) return nullptr;
    sf_frames *fr = new sf_frames();
    fr->data_folder = data_folder;
    fr->map_name = map_name;
    fr->num_poses_max = (std::size_t)std::max<int64_t>(num_poses_max, 0);
    return fr;
}

extern "C" void sf_frames_destroy(sf_frames *fr) { delete fr; }

// getMapCloud (:93-108) + mergeScansAndSave (:110-151).  The tiles are merged in sorted file
// name order (the reference uses readdir order, which is file-system dependent: a documented,
// deterministic divergence that only permutes the input of the voxel grid).
extern "C" int sf_frames_get_map_cloud(sf_frames *fr, sf_cloud *out, float voxel_size, int *loaded_cached)
{
    SF_CHECK(fr && out, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
    const std::string cached = fr->data_folder + "/" + fr->map_name + ".pcd";
    if (access(cached.c_str(), F_OK) != -1) { // cached map: loaded as is, NO voxel grid on this branch
        if (loaded_cached) *loaded_cached = 1;
        return sf_cloud_load_pcd(out, cached.c_str());
    }
    if (loaded_cached) *loaded_cached = 0;
    DIR *dir = opendir(fr->data_folder.c_str());
    SF_CHECK(dir != nullptr, SF_ERR_INVALID, "Could not open DATA directory %s", fr->data_folder.c_str());
    std::vector<std::string> names;
    while (struct dirent *ent = readdir(dir))
        if (ends_with_pcd(ent->d_name)) names.push_back(ent->d_name);
    closedir(dir);
    std::sort(names.begin(), names.end());
    std::vector<float> all, one;
    for (const std::string &n : names) {
        SF_TRY(pcd_read(fr->data_folder + "/" + n, one));
        all.insert(all.end(), one.begin(), one.end());
    }
    SF_TRY(sf_cloud_upload(out, all.data(), (int64_t)(all.size() / 3)));
    int flags = 0;
    SF_TRY(sf_cloud_voxel_downsample(out, voxel_size, SF_VOXEL_PCL, &flags));
    return sf_cloud_save_pcd(out, cached.c_str());
    });
}

// getMapTGlobal (:182-248): parse, filterBadReadings (:153-180), truncate, computeMapTGlobal
extern "C" int sf_frames_get_map_T_global(sf_frames *fr, double T[16])
{
    SF_CHECK(fr && T, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
    std::vector<double> odom = load_odometry_positions(fr->data_folder + "/odometry_positions.txt");
    std::vector<double> lla;
    std::vector<float> yaw;
    fr->altitude_table.clear();
    load_global_info(fr->data_folder + "/gps_imu_poses.txt", lla, yaw, fr->altitude_table);
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    // filterBadReadings: only applied when the two logs have the same length (:156-161)
    if (odom.size() / 3 == yaw.size()) {
        std::vector<double> odom_f, lla_f;
        std::vector<float> yaw_f;
        for (size_t i = 0; i < yaw.size(); ++i) {
            const double nrm = std::sqrt(odom[3 * i] * odom[3 * i] + odom[3 * i + 1] * odom[3 * i + 1]);
            if (nrm < 0.1 && lla[3 * i + 2] > 0) {
                odom_f.insert(odom_f.end(), odom.begin() + 3 * i, odom.begin() + 3 * i + 3);
                lla_f.insert(lla_f.end(), lla.begin() + 3 * i, lla.begin() + 3 * i + 3);
                yaw_f.push_back(yaw[i]);
            }
        }
        odom.swap(odom_f); lla.swap(lla_f); yaw.swap(yaw_f);
    }
    if (odom.empty() || yaw.empty()) {
        sf::set_error("Error: no valid odometry or global info data!");
        return SF_OK; // the reference logs and returns the identity (:191-195)
    }
    const std::size_t compute_size = std::min<std::size_t>(yaw.size(), fr->num_poses_max);
    if (compute_size == 0) { // division by zero in the reference; keep the identity
        sf::set_error("max_map_optimization_poses is 0");
        return SF_OK;
    }
    sf_fusion_map_T_global(lla.data(), yaw.data(), (int)compute_size, T);
    return SF_OK;
    });
}

extern "C" float sf_frames_get_closest_altitude(sf_frames *fr, double lat, double lon)
{
    if (!fr) return 0.0f;
    return sf_fusion_closest_altitude(fr->altitude_table.data(), (int)(fr->altitude_table.size() / 3), lat, lon);
}

extern "C" int sf_frames_altitude_table(sf_frames *fr, double *table, int64_t cap_rows, int64_t *rows)
{
    SF_CHECK(fr, SF_ERR_INVALID, "bad arguments");
    const int64_t r = (int64_t)(fr->altitude_table.size() / 3);
    if (rows) *rows = r;
    if (table) {
        SF_CHECK(cap_rows >= r, SF_ERR_INVALID, "buffer too small");
        std::memcpy(table, fr->altitude_table.data(), sizeof(double) * fr->altitude_table.size());
    }
    return SF_OK;
}

// ------------------------------------------------------------------ recorder-side writers (MapDataSaver)
// mapping/src/map_data_save_node.cpp:12-29 (folder wiped and re-created, the two text logs with their
// header lines), :61-98 (per synchronized triple: the cloud is appended to the open tile; every
// cloud_save_rate_ = 10 clouds (map_data_save_node.h:72) the tile is written as cloud_<counter>.pcd
// with pcl::io::savePCDFileBinary and cleared; one "tx ty tz" line with default ostream formatting;
// one "lat lon alt y" line with std::fixed << std::setprecision(8)), :100-112 (the open tile is
// flushed on shutdown).  Host code: it is file I/O around the path, not arithmetic on it.
struct sf_recorder {
    std::string folder;
    std::vector<float> tile;
    int cloud_counter = 0;
    int save_rate = 10;
};

namespace {
int recorder_flush(sf_recorder *r)
{
    const std::string path = r->folder + "/cloud_" + std::to_string(r->cloud_counter) + ".pcd";
    SF_TRY(pcd_write_binary(path, r->tile.data(), (int64_t)(r->tile.size() / 3)));
    r->tile.clear();
    return SF_OK;
}

int remove_tree(const std::string &path)
{
    DIR *dir = opendir(path.c_str());
    if (!dir) return errno == ENOENT ? 0 : -1;
    int rc = 0;
    while (struct dirent *ent = readdir(dir)) {
        const std::string n = ent->d_name;
        if (n == "." || n == "..") continue;
        const std::string child = path + "/" + n;
        struct stat st;
        if (lstat(child.c_str(), &st) != 0) { rc = -1; continue; }
        if (S_ISDIR(st.st_mode)) rc |= remove_tree(child);
        else if (unlink(child.c_str()) != 0) rc = -1;
    }
    closedir(dir);
    if (rmdir(path.c_str()) != 0) rc = -1;
    return rc;
}
} // namespace

extern "C" int sf_recorder_create(const char *map_data_path, sf_recorder **out)
{
    SF_CHECK(map_data_path && out && map_data_path[0], SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        const std::string folder = map_data_path;
        struct stat st;
        if (stat(folder.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) // :16-20 (the reference shells out to rm -rf)
            SF_CHECK(remove_tree(folder) == 0, SF_ERR_INVALID, "cannot clear %s", folder.c_str());
        SF_CHECK(mkdir(folder.c_str(), 0755) == 0 || errno == EEXIST, SF_ERR_INVALID, "cannot create %s", folder.c_str());
        {
            std::ofstream f(folder + "/odometry_positions.txt");
            SF_CHECK(f.is_open(), SF_ERR_INVALID, "cannot create the odometry log in %s", folder.c_str());
            f << "tx ty tz\n";
        }
        {
            std::ofstream f(folder + "/gps_imu_poses.txt");
            SF_CHECK(f.is_open(), SF_ERR_INVALID, "cannot create the GPS log in %s", folder.c_str());
            f << "lat lon alt y\n";
        }
        sf_recorder *r = new sf_recorder();
        r->folder = folder;
        *out = r;
        return SF_OK;
    });
}

extern "C" int sf_recorder_add(sf_recorder *r, const float *xyz, int64_t n, const double odom_xyz[3], double lat, double lon, double alt, double compass_yaw)
{
    SF_CHECK(r && n >= 0 && (xyz || n == 0) && odom_xyz, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        r->tile.insert(r->tile.end(), xyz, xyz + 3 * (size_t)n);
        ++r->cloud_counter;
        if (r->cloud_counter % r->save_rate == 0) SF_TRY(recorder_flush(r));
        std::ofstream fo(r->folder + "/odometry_positions.txt", std::ios::app);
        fo << odom_xyz[0] << " " << odom_xyz[1] << " " << odom_xyz[2] << std::endl;
        std::ofstream fg(r->folder + "/gps_imu_poses.txt", std::ios::app);
        fg << std::fixed << std::setprecision(8) << lat << " " << lon << " " << alt << " " << compass_yaw << std::endl;
        SF_CHECK(fo.good() && fg.good(), SF_ERR_INVALID, "write to the logs in %s failed", r->folder.c_str());
        return SF_OK;
    });
}

extern "C" int sf_recorder_shutdown(sf_recorder *r)
{
    SF_CHECK(r, SF_ERR_INVALID, "bad arguments");
    return guarded([&]() -> int {
        if (!r->tile.empty()) SF_TRY(recorder_flush(r));
        return SF_OK;
    });
}

extern "C" void sf_recorder_destroy(sf_recorder *r) { delete r; }

// map_data_save_node.cpp:38-49 (double arithmetic; the localization node's float twin is sf_fusion_compass_to_yaw)
extern "C" double sf_recorder_compass_yaw(double compass_hdg_deg)
{
    double yaw = (90.0 - compass_hdg_deg) * M_PI / 180.0;
    if (yaw > M_PI) yaw -= 2 * M_PI;
    else if (yaw < -M_PI) yaw += 2 * M_PI;
    return yaw;
}
