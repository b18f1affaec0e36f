// Text pre-step of Pocket TTS: sentence clean-up followed by SentencePiece encoding.
// Same interface as Tokenizer in gudrob/pocket-tts.cpp (include/pocket_tts/tokenizer.hpp:18-39), but
// self-contained: the .model file (a SentencePiece ModelProto) is parsed and decoded here, with no
// dependency on libsentencepiece.  Unigram and BPE models are supported.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pocket_tts {

class __attribute__((visibility("default"))) Tokenizer {
public:
    explicit Tokenizer(const std::string& modelPath);
    ~Tokenizer();
    Tokenizer(Tokenizer&&) noexcept;
    Tokenizer& operator=(Tokenizer&&) noexcept;
    Tokenizer(const Tokenizer&) = delete;
    Tokenizer& operator=(const Tokenizer&) = delete;

    // trim -> append '.' if the last character is alphanumeric -> capitalise the first letter -> ids
    std::vector<int64_t> encode(const std::string& text) const;
    // The clean-up alone (what encode() feeds to SentencePiece).
    static std::string prepareText(const std::string& text);
    // SentencePiece encoding alone, no clean-up.
    std::vector<int64_t> encodeRaw(const std::string& text) const;
    int vocabSize() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pocket_tts
