// datasource.hpp -- mirrors include/datasource.hpp:9-87 (DataElement / StereoDataElement / DataSource /
// CameraIntrinsics) and src/sources/kitti.cpp (KITTIDataSource: calib.txt -> Q, image_2 / image_3 frames).  Frames are
// read from PNG (own zlib-based reader; binary PGM/PPM accepted as well).  The ZED SDK source is out of scope.
#pragma once
#include <memory>
#include <string>

#include "image.hpp"

namespace cart {

// The Q matrix OpenCV uses to reproject disparity into 3D (datasource.hpp:11-19), row-major 4x4 float
struct CameraIntrinsics {
    float Q[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

enum DataElementType { STEREO };

class DataElement {
   public:
    explicit DataElement(DataElementType type) : type(type) {}
    virtual ~DataElement() = default;
    const DataElementType type;
};

class StereoDataElement : public DataElement {
   public:
    StereoDataElement() : DataElement(DataElementType::STEREO) {}
    StereoDataElement(image_t left, image_t right) : DataElement(DataElementType::STEREO), left(left), right(right) {}
    image_t left;
    image_t right;
};

// src/datasource.cpp:18-28: the image the per-frame modules work on (left camera)
inline image_t getReferenceImage(std::shared_ptr<DataElement> element) {
    if (element->type == DataElementType::STEREO) return std::static_pointer_cast<StereoDataElement>(element)->left;
    throw std::runtime_error("Unknown data element type");
}

class DataSource {
   public:
    explicit DataSource(Size imageSize) : imageSize(imageSize) {}
    virtual ~DataSource() = default;
    std::shared_ptr<DataElement> getNext() { return getNextInternal(); }
    virtual bool isNextReady() = 0;
    virtual bool isFinished() = 0;
    virtual DataElementType getProvidedType() = 0;
    const Size getImageSize() const { return imageSize; }
    const CameraIntrinsics getCameraIntrinsics() const { return intrinsics; }
    virtual std::string getPath() const { return std::string(); }  // directory of the sequence, if the source has one

   protected:
    virtual std::shared_ptr<DataElement> getNextInternal() = 0;
    CameraIntrinsics intrinsics;
    Size imageSize;
};

namespace sources {
// <path>/sequences/<seq>/{calib.txt, image_2/%06d.png, image_3/%06d.png} (src/sources/kitti.cpp:89-149).
// calib.txt is required like in the reference unless the frames are PGM/PPM test images.
class KITTIDataSource : public DataSource {
   public:
    // imageSize (0, 0) = the files' own size; otherwise every frame is resized to it on the device and Q is scaled
    // (kitti.hpp:11, kitti.cpp:133-148, 169-172)
    KITTIDataSource(const std::string &basePath, int sequence, Size imageSize = Size{});
    ~KITTIDataSource() override;
    bool isNextReady() override { return !isFinished(); }
    bool isFinished() override;
    DataElementType getProvidedType() override { return DataElementType::STEREO; }
    std::string getPath() const override { return dir; }

   protected:
    std::shared_ptr<DataElement> getNextInternal() override;

   private:
    class ReadAhead;
    std::string dir;
    int currentFrame = 0;
    int readAheadWorkers = 0;
    Size fileSize{};   // size of the files on disk
    std::unique_ptr<ReadAhead> readAhead;  // started by the first getNext
};
}  // namespace sources
}  // namespace cart
