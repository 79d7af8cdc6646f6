// Host-only C++ test of part 3 of the ABI through the shim classes of include/metalchat_hip.hpp:
// restates test/test_safetensor.cc ("Test write and read small model", "Test tensor link",
// "Test sharded document"), test/test_huggingface.cc:41-86 and test/test_reference.cc:17-45
// (options serializers).  Needs no GPU.  argv[1] = scratch directory.  Exit code 0 = passed.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
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
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);

    // ---- write and read small model: linear1.weight f32 [10,20], linear2.weight bf16 [3,4]
    std::vector<float> w1(10 * 20);
    for (auto& v : w1) v = dist(gen);
    std::vector<uint16_t> w2(3 * 4);
    for (auto& v : w2) {
        const float f = dist(gen);
        uint32_t u;
        std::memcpy(&u, &f, 4);
        v = (uint16_t)(u >> 16);
    }
    {
        safetensor_document out;
        out.insert("linear1.weight", "F32", {10, 20}, w1.data());
        out.insert("linear2.weight", "BF16", {3, 4}, w2.data());
        out.save(dir + "/model.st");
    }
    {
        auto in = safetensor_document::open(dir + "/model.st");
        REQUIRE(in.size() == 2);
        auto l1 = in.at("linear1.weight");
        REQUIRE(l1.dtype() == "F32" && l1.dimensions() == 2 && l1.size(0) == 10 && l1.size(1) == 20);
        REQUIRE(std::memcmp(l1.data_ptr(), w1.data(), w1.size() * 4) == 0);
        auto l2 = in.at("linear2.weight");
        REQUIRE(l2.dtype() == "BF16" && l2.numel() == 12);
        REQUIRE(std::memcmp(l2.data_ptr(), w2.data(), w2.size() * 2) == 0);
        // file order = ascending offsets = insertion order
        REQUIRE(in[0].name() == "linear1.weight" && in[1].name() == "linear2.weight");
        try {
            in.at("nope");
            REQUIRE(false);
        } catch (const std::invalid_argument&) {
        }
    }
    // ---- tensor link: output.weight shares the container of input.weight
    {
        std::vector<float> t(12, 1.5f);
        safetensor_document doc;
        doc.insert("input.weight", "F32", {3, 4}, t.data());
        doc.insert("output.weight", "input.weight");
        auto o = doc.at("output.weight");
        REQUIRE(o.size(0) == 3 && o.size(1) == 4);
        REQUIRE(o.data_ptr() == doc.at("input.weight").data_ptr());
    }
    // ---- sharded document: two files, one index
    {
        std::vector<float> t1(12, 2.0f), t2(60, 3.0f);
        safetensor_document d1, d2;
        d1.insert("tensor1", "F32", {4, 3}, t1.data());
        d1.save(dir + "/tensors-0001-of-0002.safetensors");
        d2.insert("tensor2", "F32", {10, 6}, t2.data());
        d2.save(dir + "/tensors-0002-of-0002.safetensors");
        std::ofstream idx(dir + "/tensors.safetensors.index.json");
        idx << "{\"metadata\": {}, \"weight_map\": {\"tensor1\": \"tensors-0001-of-0002.safetensors\", "
               "\"tensor2\": \"tensors-0002-of-0002.safetensors\"}}";
        idx.close();
        auto doc = safetensor_document::open_sharded(dir + "/tensors.safetensors.index.json");
        REQUIRE(doc.size() == 2);
        REQUIRE(doc.at("tensor2").size(0) == 10);
    }
    // ---- a missing / corrupt file is a runtime_error
    try {
        safetensor_document::open(dir + "/absent.safetensors");
        REQUIRE(false);
    } catch (const std::runtime_error& e) {
        REQUIRE(std::strstr(e.what(), "safetensor_document") != nullptr);
    }
    // ---- options serializers: the reference's known answers
    {
        const std::string hf = R"({"head_dim": 64, "hidden_size": 2048, "intermediate_size": 8192,
            "num_attention_heads": 32, "num_hidden_layers": 16, "num_key_value_heads": 8, "rms_norm_eps": 1e-05,
            "rope_scaling": {"factor": 32.0, "rope_type": "llama3"}, "rope_theta": 500000.0, "vocab_size": 128256})";
        const std::string meta = R"({"dim": 2048, "n_layers": 16, "n_heads": 32, "n_kv_heads": 8, "vocab_size": 128256,
            "ffn_dim_multiplier": 1.5, "multiple_of": 256, "norm_eps": 1e-05, "rope_theta": 500000.0, "use_scaled_rope": true})";
        for (int f = 0; f < 2; f++) {
            const mc_decoder_config c = load_options(f ? meta : hf, f ? MC_CKPT_META_LLAMA3 : MC_CKPT_HF_LLAMA3);
            REQUIRE(c.head_dim == 64 && c.n_layers == 16 && c.n_heads == 32 && c.n_kv_heads == 8 && c.max_seq_len == 1024);
            REQUIRE(std::fabs(c.rope_theta - 500000.0f) <= 5000.0f && std::fabs(c.norm_eps - 1e-5f) <= 1e-7f);
        }
        try {
            load_options("[1, 2", MC_CKPT_HF_LLAMA3);
            REQUIRE(false);
        } catch (const std::invalid_argument&) {
        }
    }
    std::printf("model_io ok\n");
    return 0;
}
