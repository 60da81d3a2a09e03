// datasource.hpp -- mirrors include/datasource.hpp:9-87: DataElement / StereoDataElement / DataSource.  The KITTI
// PNG loader and the ZED SDK source are out of scope this round; RawSequenceDataSource reads the same
// directory shape (image_2/%06d, image_3/%06d) from binary PGM/PPM files.
#pragma once
#include <memory>
#include <string>

#include "image.hpp"

namespace cart {

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

class DataSource {
   public:
    explicit DataSource(Size imageSize) : imageSize(imageSize) {}
    virtual ~DataSource() = default;
    std::shared_ptr<DataElement> getNext() { return getNextInternal(); }
    virtual bool isNextReady() = 0;
    virtual bool isFinished() = 0;
    virtual DataElementType getProvidedType() = 0;
    const Size getImageSize() const { return imageSize; }

   protected:
    virtual std::shared_ptr<DataElement> getNextInternal() = 0;
    Size imageSize;
};

namespace sources {
// <path>/sequences/<seq>/image_2/%06d.{pgm,ppm} and image_3/... (layout of src/sources/kitti.cpp:89-149)
class RawSequenceDataSource : public DataSource {
   public:
    RawSequenceDataSource(const std::string &basePath, int sequence);
    bool isNextReady() override { return !isFinished(); }
    bool isFinished() override;
    DataElementType getProvidedType() override { return DataElementType::STEREO; }

   protected:
    std::shared_ptr<DataElement> getNextInternal() override;

   private:
    std::string dir;
    int currentFrame = 0;
};
}  // namespace sources
}  // namespace cart
