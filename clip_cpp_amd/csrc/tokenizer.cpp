// tokenizer.cpp — clip_tokenize with bit-exact token ids (reference clip.cpp:598-679).
//
// The reference splits the text with std::regex
//   's|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+
// (ECMAScript: leftmost match, alternatives tried in order, "C" locale classes) and then looks every
// piece up greedily (longest prefix first) in the vocabulary — it is NOT byte-pair merging, does not
// lower-case, and hard-codes BOS 49406 / EOS 49407.  This file implements the same
// function with a hand-written scanner (std::regex costs ~100x more per call); equivalence with the
// regex is fuzz-tested in tests/test_host_api.py (test_tokenizer_*) against the oracle's std::regex restatement.
#include <cstdio>
#include <string>

#include "model.h"

namespace clipamd {

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }  // isspace, "C" locale
inline bool is_alpha(unsigned char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_other(unsigned char c) { return !is_space(c) && !is_alpha(c) && !is_digit(c); }

// length of the regex match that starts at s[0] (always >= 1 for a non-empty string)
size_t match_len(const char * s, size_t n) {
    const unsigned char * u = (const unsigned char *)s;
    // alternatives 1-7: contractions
    if (u[0] == '\'' && n >= 2) {
        if (u[1] == 's' || u[1] == 't') return 2;
        if (n >= 3 && u[1] == 'r' && u[2] == 'e') return 3;
        if (n >= 3 && u[1] == 'v' && u[2] == 'e') return 3;
        if (u[1] == 'm') return 2;
        if (n >= 3 && u[1] == 'l' && u[2] == 'l') return 3;
        if (u[1] == 'd') return 2;
    }
    // " ?[[:alpha:]]+" , " ?[[:digit:]]+" , " ?[^\s[:alpha:][:digit:]]+"
    const size_t lead = (u[0] == ' ' && n >= 2) ? 1 : 0;
    bool (*classes[3])(unsigned char) = {is_alpha, is_digit, is_other};
    for (auto cls : classes) {
        // with the optional space taken
        if (lead && cls(u[1])) {
            size_t i = 2;
            while (i < n && cls(u[i])) i++;
            return i;
        }
        // without it
        if (cls(u[0])) {
            size_t i = 1;
            while (i < n && cls(u[i])) i++;
            return i;
        }
    }
    // "\s+(?!\S)" then "\s+": u[0] is whitespace here
    size_t run = 1;
    while (run < n && is_space(u[run])) run++;
    if (run == n) return run;       // run reaches the end of the text
    if (run >= 2) return run - 1;   // give one back so that the look-ahead sees whitespace
    return 1;                       // single whitespace before a non-space: plain \s+
}

}  // namespace

bool tokenize_text(const clip_ctx * ctx, const char * text, std::vector<int32_t> & out) {
    out.clear();
    out.push_back(49406);  // <|startoftext|> (reference clip.cpp:637)
    const std::string str = text ? text : "";
    size_t pos = 0;
    const auto & map = ctx->token_to_id;
    std::string cand;
    while (pos < str.size()) {
        const size_t len = match_len(str.data() + pos, str.size() - pos);
        const std::string word = str.substr(pos, len);
        pos += len;
        // whole word + "</w>" first (leading space stripped)
        cand.assign(word[0] == ' ' ? word.substr(1) : word);
        cand += "</w>";
        auto wit = map.find(cand);
        if (wit != map.end()) {
            out.push_back(wit->second);
            continue;
        }
        // greedy longest match over the ORIGINAL piece (leading space included, reference clip.cpp:655-668)
        for (size_t i = 0; i < word.size();) {
            size_t j = word.size() - 1;
            if (ctx->max_token_len && j - i + 1 > ctx->max_token_len) j = i + ctx->max_token_len - 1;
            bool found = false;
            for (;; j--) {
                cand.assign(word, i, j - i + 1);
                auto it = map.find(cand);
                if (it != map.end()) {
                    out.push_back(it->second);
                    i = j + 1;
                    found = true;
                    break;
                }
                if (j == i) break;
            }
            if (!found) {
                fprintf(stderr, "%s: unknown token '%c'\n", "clip_tokenize", word[i]);
                i++;
            }
        }
    }
    out.push_back(49407);  // <|endoftext|> (reference clip.cpp:671)
    return true;
}

}  // namespace clipamd
