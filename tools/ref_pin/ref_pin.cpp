// ref_pin.cpp -- pins the oracle to the reference's own SGM dependency.  NOT part of the product and NOT buildable in
// the development container (no OpenCV): anyone with OpenCV >= 4.5 built WITH_CUDA + opencv_contrib (cudastereo,
// cudaimgproc) runs it once and commits the outputs; tests/test_ref_pin.py then compares oracle and engine with them.
//
// It makes exactly the calls the reference makes (LorgeN/CART-SLAM include/modules/disparity.hpp:26-34: createStereoSGM(minDisp,
// numDisp), setUniquenessRatio(12), setBlockSize(3); src/modules/disparity/disparity.cu:66-71: cvtColor(BGR2GRAY) x2, compute)
// on the inputs tools/ref_pin/export_inputs.py writes from the committed golden fixtures, and stores the CV_16S result.
//
//   g++ -O2 ref_pin.cpp -o ref_pin $(pkg-config --cflags --libs opencv4)
//   python export_inputs.py && ./ref_pin inputs ../../tests/golden/ref
#include <fstream>
#include <iostream>
#include <opencv2/cudaimgproc.hpp>
#include <opencv2/cudastereo.hpp>
#include <opencv2/imgcodecs.hpp>
#include <sstream>

int main(int argc, char **argv) {
    if (argc < 3) { std::cerr << "usage: ref_pin <inputs dir> <output dir>\n"; return 1; }
    const std::string in = argv[1], out = argv[2];
    std::ifstream list(in + "/cases.txt");   // one case per line: name min_disparity num_disparities paths
    if (!list) { std::cerr << "cannot read " << in << "/cases.txt (run export_inputs.py first)\n"; return 1; }
    std::string line;
    int done = 0;
    while (std::getline(list, line)) {
        std::istringstream ss(line);
        std::string name; int minDisp, numDisp, paths;
        if (!(ss >> name >> minDisp >> numDisp >> paths)) continue;
        cv::Mat l = cv::imread(in + "/" + name + "_left.png", cv::IMREAD_UNCHANGED), r = cv::imread(in + "/" + name + "_right.png", cv::IMREAD_UNCHANGED);
        if (l.empty() || r.empty()) { std::cerr << name << ": cannot read the input images\n"; return 1; }
        cv::cuda::GpuMat dl(l), dr(r), gl, gr, disp;
        if (l.channels() == 3) {   // disparity.cu:66-67
            cv::cuda::cvtColor(dl, gl, cv::COLOR_BGR2GRAY);
            cv::cuda::cvtColor(dr, gr, cv::COLOR_BGR2GRAY);
        } else { gl = dl; gr = dr; }
        // disparity.hpp:31-33 (the reference leaves P1 = 10, P2 = 120 and the mode at OpenCV's defaults; the fixtures with 8
        // paths ask for MODE_HH, the 4-path ones for the default MODE_HH4)
        cv::Ptr<cv::cuda::StereoSGM> sgm = cv::cuda::createStereoSGM(minDisp, numDisp, 10, 120, 5, paths == 8 ? cv::cuda::StereoSGM::MODE_HH : cv::cuda::StereoSGM::MODE_HH4);
        sgm->setUniquenessRatio(12);
        sgm->setBlockSize(3);
        sgm->compute(gl, gr, disp);   // disparity.cu:71
        cv::Mat h; disp.download(h);
        if (h.type() != CV_16SC1 || !h.isContinuous()) { std::cerr << name << ": unexpected output type " << h.type() << "\n"; return 1; }
        std::ofstream f(out + "/ref_disparity_" + name + ".bin", std::ios::binary);
        f.write(reinterpret_cast<const char *>(h.data), (std::streamsize)h.total() * 2);
        std::cout << name << ": " << h.cols << "x" << h.rows << " written (OpenCV " << CV_VERSION << ")\n";
        ++done;
    }
    std::ofstream(out + "/OPENCV_VERSION.txt") << CV_VERSION << "\n";
    return done ? 0 : 1;
}
