// PointsTransform: a point through a FROG transform chain, on the GPU (tools/PointsTransform.cxx).
//   PointsTransform [-p x y z] [-t transform] [-ti inverse_transform] [-o outputFileName]
// -t concatenates a transform file, -ti its inverse (reversed links, inverted matrices, Newton on the
// lattices: frog_chain_invert_links).  The outer vtkGeneralTransform is in VTK's default PreMultiply mode
// (:25-26), so of several -t/-ti the one given LAST is applied to the point FIRST.
#include "frog_chain.h"
#include "frog_host.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

extern "C" const char *frog_last_error(void);
using std::cout;
using std::endl;

int main(int argc, char *argv[])
{
    if (argc < 3) {
        std::cout << "Usage : PointsTransform [-p x y z] [-t transform] [-ti inverse_transform] [-o outputFileName]" << std::endl;
        exit(1);
    }
    std::vector<frog_transform_file *> files;
    std::vector<frog_chain_link> links;
    double *point = 0;
    int argumentsIndex = 1;
    while (argumentsIndex < argc) {
        char *key = argv[argumentsIndex];
        char *value = argumentsIndex + 1 < argc ? argv[argumentsIndex + 1] : (char *)"";
        if (strcmp(key, "-t") == 0) {
            int status = 0;
            frog_transform_file *f = frog_transform_read(value, &status);
            if (!f) { cout << "Error : cannot read transform " << value << endl; exit(1); }
            files.push_back(f);
            const uint32_t n = frog_transform_num_links(f);
            std::vector<frog_chain_link> group(frog_transform_links(f), frog_transform_links(f) + n);
            links.insert(links.begin(), group.begin(), group.end());              // PreMultiply: applied before what is there
        }
        if (strcmp(key, "-ti") == 0) {
            int status = 0;
            frog_transform_file *f = frog_transform_read(value, &status);
            if (!f) { cout << "Error : cannot read transform " << value << endl; exit(1); }
            files.push_back(f);
            const uint32_t n = frog_transform_num_links(f);
            std::vector<frog_chain_link> group(n);
            if (frog_chain_invert_links(frog_transform_links(f), n, group.data())) { cout << "Error : " << frog_last_error() << endl; exit(1); }
            links.insert(links.begin(), group.begin(), group.end());
        }
        if (strcmp(key, "-p") == 0) {
            if (argumentsIndex + 3 >= argc) { cout << "Error : -p needs three values" << endl; exit(1); }
            point = new double[3];
            for (int i = 0; i < 3; i++) point[i] = atof(argv[argumentsIndex + 1 + i]);
            argumentsIndex += 2;
        }
        argumentsIndex += 2;
    }
    if (point) {
        double newPoint[3];
        cout << "Input point : " << point[0] << " " << point[1] << " " << point[2] << endl;
        frog_chain *c = nullptr;
        if (frog_chain_create(links.data(), (uint32_t)links.size(), 0, &c) || frog_chain_apply(c, point, newPoint, 1)) {
            cout << "Error : " << frog_last_error() << endl;
            exit(1);
        }
        cout << "Output point : " << newPoint[0] << " " << newPoint[1] << " " << newPoint[2] << endl;
        frog_chain_destroy(c);
        delete[] point;
    }
    for (auto *f : files) frog_transform_free(f);
    return 0;
}
