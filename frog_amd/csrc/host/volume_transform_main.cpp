// VolumeTransform: resample a volume on the grid of a reference volume through a FROG transform chain, on
// the GPU (tools/VolumeTransform.cxx).
//   VolumeTransform source reference [-t transform] [-ti inverse_transform] [-i interpolation] [-o outputFileName]
//                   [-rx reverseX] [-b backgroundLevel]
// vtkImageReslice needs the map from the OUTPUT grid to the source: -t (a registration transform of the
// source, source -> common space) is therefore inverted (:55-57), -ti is taken as it is (:60-62).  The outer
// vtkGeneralTransform is in VTK's default PreMultiply mode: of several -t/-ti the last one acts first.
// Volumes: NIfTI-1 and MetaImage (frog_host.h); the reference reads more formats through
// vtkRobustImageReader (absent submodule).  New: -dev <n> selects the HIP device.
#include "frog_chain.h"
#include "frog_host.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

extern "C" const char *frog_last_error(void);

int main(int argc, char *argv[])
{
    if (argc < 3) {
        std::cout << "Usage : VolumeTransform source reference [-t transform] [-ti inverse_transform] [-i interpolation] [-o outputFileName] [-rx reverseX]" << std::endl;
        exit(1);
    }
    int reverseX = 0, device = 0;
    char *outputFile = 0;
    int interpolation = 1;              // linear
    bool backGroundSet = false;
    float backGroundLevel = 0;
    std::vector<frog_transform_file *> files;
    std::vector<frog_chain_link> links;
    auto die = [](const std::string &what) { std::cout << "Error : " << what << std::endl; exit(1); };

    int argumentsIndex = 3;
    while (argumentsIndex < argc) {
        char *key = argv[argumentsIndex];
        char *value = argumentsIndex + 1 < argc ? argv[argumentsIndex + 1] : (char *)"";
        if (strcmp(key, "-b") == 0) { backGroundLevel = atof(value); backGroundSet = true; }
        if (strcmp(key, "-t") == 0 || strcmp(key, "-ti") == 0) {
            int status = 0;
            frog_transform_file *f = frog_transform_read(value, &status);
            if (!f) die(std::string("cannot read transform ") + value);
            files.push_back(f);
            const uint32_t n = frog_transform_num_links(f);
            std::vector<frog_chain_link> group(frog_transform_links(f), frog_transform_links(f) + n);
            if (strcmp(key, "-t") == 0 && frog_chain_invert_links(frog_transform_links(f), n, group.data())) die(frog_last_error());
            links.insert(links.begin(), group.begin(), group.end());
        }
        if (strcmp(key, "-o") == 0) outputFile = value;
        if (strcmp(key, "-i") == 0) interpolation = atoi(value);
        if (strcmp(key, "-rx") == 0) reverseX = atoi(value);
        if (strcmp(key, "-dev") == 0) device = atoi(value);
        argumentsIndex += 2;
    }

    using clk = std::chrono::steady_clock;
    frog_volume_file *volumes[2];
    frog_volume views[2];
    // the two volumes are read side by side (inflating a compressed 256^3 volume is 0.16 s of one thread) while the HIP runtime
    // comes up on a third thread (0.05-0.2 s, otherwise inside frog_chain_create below); messages in upstream's order afterwards
    static std::thread warm;
    warm = std::thread([device] { (void)frog_device_warm(device); });
    std::atexit([] { if (warm.joinable()) warm.join(); });
    double loadSeconds[2] = { 0, 0 };
    volumes[0] = volumes[1] = nullptr;
    {
        auto load = [&](int i) {
            auto t0 = clk::now();
            int status = 0;
            volumes[i] = frog_volume_read(argv[i + 1], &status);
            loadSeconds[i] = std::chrono::duration<double>(clk::now() - t0).count();
        };
        std::thread other(load, 0);
        load(1);
        other.join();
    }
    for (int i = 0; i < 2; i++) {
        std::cout << "load : " << argv[i + 1] << std::endl;
        if (!volumes[i]) die(std::string("cannot read volume ") + argv[i + 1]);
        frog_volume_view(volumes[i], &views[i]);
        std::cout << "Image loaded in " << loadSeconds[i] << "s" << std::endl;
    }
    const frog_volume &src = views[0], &ref = views[1];
    double bounds[6], center[3], transformedCenter[3];
    for (int k = 0; k < 3; k++) {                   // vtkImageData::GetBounds: first and last voxel centre
        const double a = src.origin[k], b = src.origin[k] + (src.dims[k] - 1) * src.spacing[k];
        bounds[2 * k] = a < b ? a : b; bounds[2 * k + 1] = a < b ? b : a;
        center[k] = 0.5 * (bounds[2 * k] + bounds[2 * k + 1]);
    }
    std::cout << "image bounds :";
    for (int i = 0; i < 6; i++) std::cout << bounds[i] << " ";
    std::cout << std::endl;
    std::cout << "center :" << center[0] << " " << center[1] << " " << center[2] << std::endl;

    frog_chain *chain = nullptr;
    if (frog_chain_create(links.data(), (uint32_t)links.size(), device, &chain)) die(frog_last_error());
    if (frog_chain_apply(chain, center, transformedCenter, 1)) die(frog_last_error());
    std::cout << "transformed center :" << transformedCenter[0] << " " << transformedCenter[1] << " " << transformedCenter[2] << std::endl;

    double valueRange[2] = { 0, 0 };
    frog_volume_range(&src, &valueRange[0], &valueRange[1]);

    frog_volume out = ref;                          // output grid = the reference volume's (:121-123), scalars as the source
    out.dtype = src.dtype;
    const size_t nOut = (size_t)out.dims[0] * out.dims[1] * out.dims[2];
    std::vector<unsigned char> data(nOut * frog_volume_voxel_bytes(out.dtype));
    out.data = data.data();
    auto t0 = clk::now();
    if (frog_chain_reslice(chain, &src, &out, interpolation, backGroundSet ? (double)backGroundLevel : valueRange[0])) die(frog_last_error());
    std::cout << "Transform computed in " << std::chrono::duration<double>(clk::now() - t0).count() << "s" << std::endl;

    if (reverseX) {                                 // vtkImageFlip along x (:190-195): voxels mirrored, geometry kept
        const size_t vb = frog_volume_voxel_bytes(out.dtype), nx = out.dims[0];
        std::vector<unsigned char> tmp(vb);
        for (size_t row = 0; row < nOut / nx; row++) {
            unsigned char *r = data.data() + row * nx * vb;
            for (size_t a = 0, b = nx - 1; a < b; a++, b--) {
                std::memcpy(tmp.data(), r + a * vb, vb); std::memcpy(r + a * vb, r + b * vb, vb); std::memcpy(r + b * vb, tmp.data(), vb);
            }
        }
    }
    t0 = clk::now();
    const char *name = outputFile ? outputFile : "output.mhd";
    if (frog_volume_write(name, &out)) {
        std::cout << "not able to write  " << name << std::endl;          // :176 (upstream's message says "read")
        return 1;
    }
    std::cout << "File written in " << std::chrono::duration<double>(clk::now() - t0).count() << "s" << std::endl;

    frog_chain_destroy(chain);
    for (auto *f : files) frog_transform_free(f);
    for (auto *v : volumes) frog_volume_free(v);
    return 0;
}
