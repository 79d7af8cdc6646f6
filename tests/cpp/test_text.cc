// Host-only C++ test of part 4 of the ABI through the shim classes of include/metalchat_hip.hpp:
// the reference's GPT-2 codec known answers (test/test_bpe.cc:29-55), a byte_pair_encoder built by
// hand, the reference's exception types and texts (bpe.h:304-342), a tiktoken file with the llama3
// control tokens behind it, and interpreter::write framing.  Needs no GPU.  argv[1] = scratch
// directory.  Exit code 0 = passed.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "metalchat_hip.hpp"

using namespace metalchat::hip;

#define REQUIRE(cond)                                                    \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int
main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];

    text::gpt2_codec codec;
    REQUIRE(codec.encode("    Hello  \x80") == "ĠĠĠĠHelloĠĠĢ");
    REQUIRE(codec.decode("ĠĠĠĠHelloĠĠĢ") == "    Hello  \x80");
    REQUIRE(codec.encode(" استاندارد") == "ĠØ§Ø³ØªØ§ÙĨØ¯Ø§Ø±Ø¯");
    REQUIRE(codec.decode("ĠØ§Ø³ØªØ§ÙĨØ¯Ø§Ø±Ø¯") == " استاندارد");

    text::bpe t("[a-z]+|.");
    t.insert("a", 0);
    t.insert("b", 1);
    t.insert("ab", 2);
    REQUIRE(t.size() == 3);
    REQUIRE((t.encode("ab b") == std::vector<int32_t>{2, 1})); // " " is no token and has no pair to merge into
    REQUIRE((t.encode("ba") == std::vector<int32_t>{1}));       // bpe.h:137-145: the unmerged last byte is dropped
    REQUIRE(t.decode(2) == "ab");
    bool thrown = false;
    try {
        t.decode(77);
    } catch (const std::runtime_error& e) {
        thrown = std::string(e.what()) == "byte_pair_encoder: unable to decode id '77'";
    }
    REQUIRE(thrown);
    thrown = false;
    try {
        t.encode(MC_TOKEN_END_TURN);
    } catch (const std::invalid_argument& e) {
        thrown = std::string(e.what()) == "byte_pair_encoder: unknown control token '256'";
    }
    REQUIRE(thrown);
    thrown = false;
    try {
        text::bpe bad("(unclosed");
    } catch (const std::invalid_argument& e) {
        thrown = std::string(e.what()).rfind("regexp: invalid regular expression", 0) == 0;
    }
    REQUIRE(thrown);

    // "YQ== 0" ... : a, b, ab, "\n\n" in tiktoken form, the llama3 control tokens follow at 4 .. 14
    const std::string path = dir + "/tokenizer.model";
    {
        std::ofstream f(path);
        f << "YQ== 0\nYg== 1\nYWI= 2\nCgo= 3\n";
    }
    text::bpe m = text::bpe::load_tiktoken(path);
    REQUIRE(m.size() == 15);
    REQUIRE(m.encode(MC_TOKEN_BEGIN_TEXT) == 4);
    REQUIRE(m.encode(MC_TOKEN_END_TEXT) == 5);
    REQUIRE(m.encode(MC_TOKEN_RESERVED) == 9); // the last reserved token inserted (src/reference.cc:121)
    REQUIRE(m.encode(MC_TOKEN_BEGIN_HEADER) == 10);
    REQUIRE(m.encode(MC_TOKEN_END_HEADER) == 11);
    REQUIRE(m.encode(MC_TOKEN_END_TURN) == 13);
    REQUIRE(m.decode(5) == "<|end_of_text|>");

    // interpreter::write: begin_text | header(role) | content | end_turn (src/interpreter.cc:116-136)
    interpreter it(nullptr, m);
    it.declare_variable("who", "ab");
    it.write(basic_message("ab", "{{ who }}"));
    REQUIRE(it.start_pos() == 0);
    thrown = false;
    try {
        it.read();
    } catch (const std::invalid_argument&) {
        thrown = true; // no decoder behind this interpreter
    }
    REQUIRE(thrown);
    std::printf("text ok\n");
    return 0;
}
