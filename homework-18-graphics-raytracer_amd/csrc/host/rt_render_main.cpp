/*
 * rt_render — command-line driver: the reference's main() (src/main.rs:809-1175)
 * with the two rayon render loops replaced by calls across the C ABI.
 *
 *   rt_render [--width W] [--height H] [--depth D] [--obj dodecahedron.obj] [--out out.png] [--epochs N]
 *             [--focus F] [--blur B] [--scene in.rtscene] [--save-scene out.rtscene] [--devices 0,1,...]
 *
 * Defaults are the reference's literals: 1280x960, depth 5 (main.rs:1084-1085, 1098).  --epochs N (default 0; the
 * reference: 100, main.rs:1129) continues as main() does: N epochs of the depth-of-field pass (shoot_focus(focus, blur),
 * defaults 3.0 / 0.04 = main.rs:1147-1148; depth 5) added to the normalised image, post_process and a rewrite of the PNG
 * after every epoch (main.rs:1129-1174).  `--epochs 7` reproduces report/out.png, `--epochs 7 --blur 0.02`
 * report/out_small_blur.png (tests/test_gpu_reference_pins.py).
 * --scene renders a scene file (rt_world_load_scene; its camera if it carries one, else main()'s) instead of the literal
 * scene; --save-scene writes the scene in use (with the camera) as such a file before rendering.
 * --devices renders on several GPUs from this one process (rt_multi_*: interleaved row bands, one per list entry; an index may
 * repeat); without it the current device renders everything.
 * Host keeps: scene build + OBJ import, post_process, sRGB/u8 encode, PNG write.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/rt_amd.h"
#include "../../../include/rt_host.h"

int main(int argc, char **argv) {
    uint32_t width = 1280, height = 960;
    int32_t depth = 5;
    const char *obj = "dodecahedron.obj";
    const char *out = "./out.png";
    const char *scene_in = nullptr, *scene_out = nullptr;
    std::vector<int> devices;
    int epochs = 0;
    float focus = 3.0f, blur = 0.04f; /* main.rs:1147-1148 */
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--width")) width = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--height")) height = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--depth")) depth = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--obj")) obj = argv[i + 1];
        else if (!strcmp(argv[i], "--out")) out = argv[i + 1];
        else if (!strcmp(argv[i], "--epochs")) epochs = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--scene")) scene_in = argv[i + 1];
        else if (!strcmp(argv[i], "--save-scene")) scene_out = argv[i + 1];
        else if (!strcmp(argv[i], "--devices")) {
            for (const char *c = argv[i + 1]; *c;) {
                devices.push_back(atoi(c));
                while (*c && *c != ',') ++c;
                if (*c == ',') ++c;
            }
        }
        else if (!strcmp(argv[i], "--focus")) focus = strtof(argv[i + 1], nullptr);
        else if (!strcmp(argv[i], "--blur")) blur = strtof(argv[i + 1], nullptr);
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    rt_world *world = rt_world_new();
    rt_camera camera;
    rt_reference_camera(&camera);
    int file_has_camera = 0;
    if (!world || (scene_in ? rt_world_load_scene(world, scene_in, &camera, &file_has_camera) : rt_world_build_reference_scene(world, obj)) != RT_OK) {
        fprintf(stderr, "scene build failed: %s\n", rt_host_last_error());
        return 1;
    }
    if (scene_out && rt_world_save_scene(world, &camera, scene_out) != RT_OK) {
        fprintf(stderr, "rt_world_save_scene failed: %s\n", rt_host_last_error());
        return 1;
    }
    rt_scene_desc desc;
    rt_world_desc(world, &desc);
    rt_frame frame;
    rt_frame_full(width, height, depth, &frame);

    rt_scene *scene = nullptr;
    rt_multi *multi = nullptr;
    if (!devices.empty()) {
        if (rt_multi_create(&desc, devices.data(), (int)devices.size(), &multi) != RT_OK) {
            fprintf(stderr, "rt_multi_create failed: %s\n", rt_last_error());
            return 1;
        }
    } else if (rt_scene_create(&desc, &scene) != RT_OK) {
        fprintf(stderr, "rt_scene_create failed: %s\n", rt_last_error());
        return 1;
    }
    std::vector<float> img((size_t)width * height * 3);
    unsigned long long casts = 0;
    auto t0 = std::chrono::steady_clock::now();
    if ((multi ? rt_multi_render_whitted_host(multi, &camera, &frame, img.data(), &casts)
               : rt_render_whitted_host(scene, &camera, &frame, img.data(), &casts)) != RT_OK) {
        fprintf(stderr, "rt_render_whitted_host failed: %s\n", rt_last_error());
        return 1;
    }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    /* the reference prints pixels as "rays" and divides by whole milliseconds (panics under 1 ms, main.rs:1111) */
    const unsigned long long pixels = (unsigned long long)width * height;
    printf("%llu rays in %.3f ms (%.0f rays/s); %llu casts (%.2f per pixel, %.1f Mcasts/s incl. copies)\n", pixels, ms,
           ms > 0 ? pixels * 1000.0 / ms : 0.0, casts, (double)casts / (double)pixels, ms > 0 ? casts / ms / 1000.0 : 0.0);

    rt_post_process(img.data(), (size_t)width * height);
    std::vector<uint8_t> rgb8(img.size());
    rt_encode_srgb8(img.data(), img.size(), rgb8.data());
    if (rt_write_png(out, rgb8.data(), width, height) != RT_OK) {
        fprintf(stderr, "rt_write_png failed: %s\n", rt_host_last_error());
        return 1;
    }
    if (epochs > 0) {
        rt_rng *rng = nullptr; /* main.rs:1117-1127 (rt_multi creates its generators on the first epoch) */
        if (!multi && rt_rng_create(&frame, &rng) != RT_OK) {
            fprintf(stderr, "rt_rng_create failed: %s\n", rt_last_error());
            return 1;
        }
        for (int i = 0; i < epochs; ++i) {
            t0 = std::chrono::steady_clock::now();
            if ((multi ? rt_multi_render_distributed_host(multi, &camera, &frame, focus, blur, 1, img.data(), &casts)
                       : rt_render_distributed_host(scene, &camera, &frame, focus, blur, rng, 1, img.data(), &casts)) != RT_OK) {
                fprintf(stderr, "rt_render_distributed_host failed: %s\n", rt_last_error());
                return 1;
            }
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%llu rays in %.3f ms (%.0f rays/s); %llu casts\n", pixels, ms, ms > 0 ? pixels * 1000.0 / ms : 0.0, casts);
            rt_post_process(img.data(), (size_t)width * height); /* main.rs:1171: img itself is renormalised and kept */
            rt_encode_srgb8(img.data(), img.size(), rgb8.data());
            if (rt_write_png(out, rgb8.data(), width, height) != RT_OK) {
                fprintf(stderr, "rt_write_png failed: %s\n", rt_host_last_error());
                return 1;
            }
        }
        rt_rng_destroy(rng);
    }
    rt_multi_destroy(multi);
    rt_scene_destroy(scene);
    rt_world_free(world);
    return 0;
}
