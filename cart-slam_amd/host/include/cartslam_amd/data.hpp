// data.hpp -- per-frame blackboard. Mirrors include/utils/data.hpp:11-79 + src/utils/data.cpp:17-56 of the reference
// (DataContainer: mutex + condition variable, string keys -> shared_ptr<void>, 20 s wait timeout).
#pragma once
#include <chrono>
#include <condition_variable>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#define CARTSLAM_WAIT_FOR_DATA_TIMEOUT 20

namespace cart {
typedef std::pair<std::string, std::shared_ptr<void>> system_data_pair_t;
typedef std::vector<system_data_pair_t> system_data_t;

class DataNotAvailableException : public std::exception {
   public:
    explicit DataNotAvailableException(const std::string &key) : key("Missing key \"" + key + "\"") {}
    const char *what() const noexcept override { return key.c_str(); }

   private:
    const std::string key;
};

class DataContainer {
   public:
    virtual ~DataContainer() = default;

    bool hasData(const std::string &key) {
        std::unique_lock<std::mutex> lock(dataMutex);
        return data.count(key) != 0;
    }

    template <typename T>
    std::shared_ptr<T> getData(const std::string &key) {
        std::unique_lock<std::mutex> lock(dataMutex);
        auto it = data.find(key);
        if (it == data.end()) throw std::invalid_argument("Could not find key \"" + key + "\"");
        return std::static_pointer_cast<T>(it->second);
    }

    // blocks until every key exists; throws DataNotAvailableException after CARTSLAM_WAIT_FOR_DATA_TIMEOUT seconds
    void waitForData(const std::vector<std::string> &keys) {
        std::unique_lock<std::mutex> lock(dataMutex);
        for (const auto &key : keys) {
            if (!dataCondition.wait_for(lock, std::chrono::seconds(CARTSLAM_WAIT_FOR_DATA_TIMEOUT), [&] { return data.count(key) != 0; }))
                throw DataNotAvailableException(key);
        }
    }

    void insertData(const system_data_pair_t &entry) {
        {
            std::unique_lock<std::mutex> lock(dataMutex);
            data[entry.first] = entry.second;
        }
        dataCondition.notify_all();
    }

   private:
    std::map<std::string, std::shared_ptr<void>> data;
    std::mutex dataMutex;
    std::condition_variable dataCondition;
};
}  // namespace cart

// include/utils/modules.hpp:5-9
#define MODULE_NO_RETURN_VALUE (std::vector<cart::system_data_pair_t>{})
#define MODULE_RETURN(key, value) (std::vector<cart::system_data_pair_t>{std::make_pair(std::string(key), std::shared_ptr<void>(value))})
#define MODULE_RETURN_ALL(...) (std::vector<cart::system_data_pair_t>{__VA_ARGS__})
#define MODULE_MAKE_PAIR(key, valueType, ...) std::make_pair(std::string(key), std::shared_ptr<void>(std::make_shared<valueType>(__VA_ARGS__)))
#define MODULE_RETURN_SHARED(key, valueType, ...) (std::vector<cart::system_data_pair_t>{MODULE_MAKE_PAIR(key, valueType, __VA_ARGS__)})
