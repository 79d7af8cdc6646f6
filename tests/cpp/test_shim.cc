// Drives the reference-named kernels through the C++ shim (include/metalchat_hip.hpp) the way the
// reference's kernel wrappers drive Metal (include/metalchat/kernel.h:166-298): a task object with
// encode(hardware_function_encoder) pushed onto the accelerator's kernel_thread.
// Restates test/test_kernel_thread.cc:16-40 (three chained adds on ones == 8) and
// test/test_accelerator.cc:15-21 (missing library -> runtime_error), plus the validation errors of
// include/metalchat/kernel.h:119-141.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>
#include <vector>

#include "metalchat_hip.hpp"

using namespace metalchat::hip;

struct add_task {
    basic_kernel kernel;
    tensor_layout<2> layout;
    shared_buffer out, a, b;
    dim3 grid, thread;

    void
    encode(hardware_function_encoder encoder)
    {
        encoder.initialize(kernel.name(), kernel.get_hip_kernel());
        // output tensor first, then the inputs: layout by value, then the buffer
        encoder.encode(layout);
        encoder.encode(out, 0);
        encoder.encode(layout);
        encoder.encode(a, 0);
        encoder.encode(layout);
        encoder.encode(b, 0);
        encoder.dispatch(grid, thread);
    }
};

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
    // wrong shader library -> std::runtime_error("hip: library not found")
    try {
        hardware_accelerator bad("some/nonexisting/file", 1);
        REQUIRE(false);
    } catch (const std::runtime_error& e) {
        REQUIRE(std::strstr(e.what(), "library not found") != nullptr);
    }

    hardware_accelerator gpu0(argv[1], /*thread_capacity=*/2);
    // unknown function -> std::invalid_argument
    try {
        gpu0.load("no_such_kernel_float");
        REQUIRE(false);
    } catch (const std::invalid_argument& e) {
        REQUIRE(std::strstr(e.what(), "not found in a shader library") != nullptr);
    }

    const std::size_t rows = 12, dim = 15, n = rows * dim;
    std::vector<float> ones(n, 1.0f);
    auto dev = gpu0.get_hip_device();
    auto kernel = gpu0.load("add", "float");
    auto [grid, thread] = make_kernel_grid_2d(rows, dim, kernel.max_threads_per_threadgroup());
    tensor_layout<2> l{{(uint32_t)rows, (uint32_t)dim}, {(uint32_t)dim, 1}, {0, 0}};

    shared_buffer cur = make_buffer(dev, ones.data(), n * sizeof(float));
    std::shared_future<void> last;
    for (int i = 0; i < 3; i++) { // capacity 2: the third push lands in a new kernel_thread
        shared_buffer out = make_buffer(dev, n * sizeof(float));
        add_task task{kernel, l, out, cur, cur, grid, thread};
        last = gpu0.get_this_thread()->push(task);
        cur = out;
    }
    gpu0.get_this_thread()->make_ready_at_thread_exit();
    last.get();
    std::vector<float> result(n);
    check(mc_queue_wait(gpu0.queue().get()));
    check(mc_buffer_download(cur.get(), 0, result.data(), n * sizeof(float)));
    for (float v : result) REQUIRE(v == 8.0f);

    // group larger than the pipeline maximum -> std::invalid_argument
    try {
        add_task task{kernel, l, cur, cur, cur, dim3(4096), dim3(2048)};
        gpu0.get_this_thread()->push(task);
        REQUIRE(false);
    } catch (const std::invalid_argument& e) {
        REQUIRE(std::strstr(e.what(), "exceeds maximum number of threads") != nullptr);
    }
    std::printf("shim ok on '%s' (max buffer %zu MiB)\n", gpu0.name().c_str(), gpu0.max_buffer_size() >> 20);
    return 0;
}
