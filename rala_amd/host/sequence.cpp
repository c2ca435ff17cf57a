#include "sequence.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace rala {

// (the factory's two fatal checks and their message texts: reference src/sequence.cpp:15-22)
std::unique_ptr<Sequence> createSequence(const std::string& name, const std::string& data) {
    const char* missing = name.empty() ? "name" : data.empty() ? "data" : nullptr;
    if (missing) {
        fprintf(stderr, "[rala::createSequence] error: empty %s!\n", missing);
        exit(1);
    }
    return std::unique_ptr<Sequence>(new Sequence(name, data));
}

void Sequence::trim(uint32_t begin, uint32_t end) {
    data_ = data_.substr(begin, end - begin);
    if (!reverse_complement_.empty()) create_reverse_complement();
}

void Sequence::create_reverse_complement() {
    reverse_complement_.assign(data_.rbegin(), data_.rend());
    for (auto& c : reverse_complement_) {
        switch (c) {
            case 'A': c = 'T'; break;
            case 'T': c = 'A'; break;
            case 'C': c = 'G'; break;
            case 'G': c = 'C'; break;
            default: break;
        }
    }
}

}  // namespace rala
