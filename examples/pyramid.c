/* A plain-C caller of librapier_hip.so: builds the BASELINE plumbing scene (one 10-base pyramid of 55 cubes on a slab), steps it
 * 300 times on the GPU and prints the apex cube's height.  Shows the calling convention of include/rapier_hip.h from C:
 *
 *   gcc -std=c99 -Iinclude examples/pyramid.c -Lrapier_amd -lrapier_hip -Wl,-rpath,$PWD/rapier_amd -o pyramid && ./pyramid
 *
 * (needs an MI355X at run time: there is no CPU fallback — rp_world_create fails without a HIP device). */
#include "rapier_hip.h"
#include <stdio.h>
#include <string.h>

static int check(rp_world *w, int32_t st, const char *what) {
    if (st == RP_OK) return 0;
    fprintf(stderr, "%s failed (%d): %s\n", what, (int)st, w ? rp_last_error(w) : "no world");
    return 1;
}

int main(void) {
    rp_integration_params params;
    rp_default_params(&params);                       /* IntegrationParameters::default() */
    const float gravity[3] = {0.0f, -10.0f, 0.0f};
    rp_world *w = NULL;
    if (check(NULL, rp_world_create(&params, gravity, 0, &w), "rp_world_create")) return 1;

    /* ground: RigidBodyBuilder::fixed().translation(0, -1, 0) + ColliderBuilder::cuboid(5.5, 1, 5.5) */
    rp_body_desc body; memset(&body, 0, sizeof body);
    body.body_type = RP_BODY_FIXED; body.translation[1] = -1.0f; body.rotation[3] = 1.0f; body.gravity_scale = 1.0f; body.gyroscopic = 1;
    uint64_t ground = 0, handle = 0, apex = 0;
    if (check(w, rp_bodies_insert(w, 1, &body, &ground), "rp_bodies_insert")) return 1;
    rp_collider_desc col; memset(&col, 0, sizeof col);
    col.shape = RP_SHAPE_CUBOID; col.half_extents[0] = 5.5f; col.half_extents[1] = 1.0f; col.half_extents[2] = 5.5f;
    col.rotation[3] = 1.0f; col.density = 1.0f; col.friction = 0.5f; col.collision_memberships = 0xffffffffu; col.collision_filter = 0xffffffffu;
    if (check(w, rp_colliders_insert(w, 1, &col, &ground, &handle), "rp_colliders_insert")) return 1;

    /* the pyramid of examples3d/b3d_many_pyramids.rs:8-29 (one pyramid, base 10, extent 0.5, density 100) */
    col.half_extents[0] = col.half_extents[1] = col.half_extents[2] = 0.5f; col.density = 100.0f;
    body.body_type = RP_BODY_DYNAMIC;
    for (int i = 0; i < 10; ++i)
        for (int j = i; j < 10; ++j) {
            body.translation[0] = (float)(i + 1) * 0.5f + 2.0f * (float)(j - i) * 0.5f - 4.5f - 0.5f;
            body.translation[1] = (2.0f * (float)i + 1.0f) * 0.5f;
            body.translation[2] = -4.5f;
            if (check(w, rp_bodies_insert(w, 1, &body, &handle), "rp_bodies_insert")) return 1;
            if (i == 9) apex = handle;
            uint64_t ch;
            if (check(w, rp_colliders_insert(w, 1, &col, &handle, &ch), "rp_colliders_insert")) return 1;
        }

    if (check(w, rp_step(w, 300), "rp_step") || check(w, rp_sync(w), "rp_sync")) return 1;   /* PhysicsWorld::step() x 300 */
    float pos7[7], vel6[6];
    if (check(w, rp_bodies_read(w, 1, &apex, pos7, vel6), "rp_bodies_read")) return 1;
    rp_counters c;
    if (check(w, rp_counters_read(w, &c), "rp_counters_read")) return 1;
    printf("apex cube at y = %.4f (rests at 9.5), %d solver manifolds, %d colours\n", pos7[1], (int)c.num_manifolds, (int)c.num_colors);
    rp_world_destroy(w);
    return 0;
}
