// Test shim: the product's host-side polyhedron code (rapier_amd/csrc/rp_polyhedron.h is plain C++) behind a C interface, so that the
// CPU suite can compare it with the oracle's construction and with Qhull without a GPU.  Built by tests/test_polyhedron_host.py with g++.
#include "../rapier_amd/csrc/rp_polyhedron.h"
extern "C" int shim_hull(int n, const float *xyz, unsigned *tris_out, int cap) {
    std::vector<uint32_t> t;
    if (!rp_poly::convex_hull(n, xyz, t)) return -1;
    if ((int)t.size() > cap) return -2;
    memcpy(tris_out, t.data(), t.size() * sizeof(uint32_t));
    return (int)t.size() / 3;
}
// counts = {vertices, faces, loop entries, edges}; arrays sized by the caller (256 vertices, 512 faces, 2048 loop entries at most)
extern "C" int shim_build(int n, const float *xyz, int nt, const unsigned *tris, int counts[4], float *pts, float *fn, int *ff, int *fc, int *lv, int *le, float props[20]) {
    HostPolyhedron P;
    if (!rp_poly::build(P, n, xyz, nt, tris)) return -1;
    counts[0] = P.nv(); counts[1] = P.nf(); counts[2] = (int)P.loop_v.size(); counts[3] = P.ne;
    memcpy(pts, P.pts.data(), P.pts.size() * 4); memcpy(fn, P.fnormal.data(), P.fnormal.size() * 4);
    memcpy(ff, P.ffirst.data(), P.ffirst.size() * 4); memcpy(fc, P.fcount.data(), P.fcount.size() * 4);
    memcpy(lv, P.loop_v.data(), P.loop_v.size() * 4); memcpy(le, P.loop_e.data(), P.loop_e.size() * 4);
    float *o = props;
    o[0] = P.centre[0]; o[1] = P.centre[1]; o[2] = P.centre[2]; o[3] = P.half[0]; o[4] = P.half[1]; o[5] = P.half[2]; o[6] = P.origin_radius;
    o[7] = P.sphere_centre[0]; o[8] = P.sphere_centre[1]; o[9] = P.sphere_centre[2]; o[10] = P.sphere_radius;
    o[11] = P.volume; o[12] = P.com[0]; o[13] = P.com[1]; o[14] = P.com[2];
    o[15] = P.inertia[0][0]; o[16] = P.inertia[1][1]; o[17] = P.inertia[2][2]; o[18] = P.inertia[0][1]; o[19] = P.inertia[0][2];
    return 0;
}
