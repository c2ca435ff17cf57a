/*!
 * @file sequence.hpp
 *
 * @brief Sequence class (interface of rvaser/rala src/sequence.hpp:25-67)
 */

#pragma once

#include <stdint.h>
#include <memory>
#include <string>

namespace rala {

class Sequence;
std::unique_ptr<Sequence> createSequence(const std::string& name, const std::string& data);

class Sequence {
public:
    ~Sequence() {}

    const std::string& name() const { return name_; }
    const std::string& data() const { return data_; }

    const std::string& reverse_complement() {
        if (reverse_complement_.size() != data_.size()) create_reverse_complement();
        return reverse_complement_;
    }

    /*! @brief keeps data_[begin, end) (reference src/sequence.cpp:39-44) */
    void trim(uint32_t begin, uint32_t end);

    friend std::unique_ptr<Sequence> createSequence(const std::string& name, const std::string& data);

private:
    Sequence(const std::string& name, const std::string& data) : name_(name), data_(data), reverse_complement_() {}
    Sequence(const Sequence&) = delete;
    const Sequence& operator=(const Sequence&) = delete;

    void create_reverse_complement();

    std::string name_;
    std::string data_;
    std::string reverse_complement_;
};

}  // namespace rala
