// tests/cpp/par_build_check.cc — the generic host builder of include/nanort.h on a user primitive (spheres through
// plain functors, as examples/particle_primitive does): prints a fingerprint of the node array + index permutation, the
// statistics and the build time.  tests/test_host_header.py compiles it serially, with OpenMP and with std::thread and
// asserts that all three produce the same tree (reference: nanort.h:2018-2117 builds in parallel under the same macros).
#include <cstdio>
#include <cstring>
#include <chrono>
#include <random>
#include "nanort.h"
struct SphereGeom { const float* c; const float* r; 
  void BoundingBox(nanort::real3<float>* mn, nanort::real3<float>* mx, unsigned i) const { for(int k=0;k<3;k++){(*mn)[k]=c[3*i+k]-r[i];(*mx)[k]=c[3*i+k]+r[i];} }
  void BoundingBoxAndCenter(nanort::real3<float>* mn, nanort::real3<float>* mx, nanort::real3<float>* ce, unsigned i) const { BoundingBox(mn,mx,i); for(int k=0;k<3;k++)(*ce)[k]=c[3*i+k]; } };
struct SpherePred { const float* c; int axis; float pos; SpherePred(const float*c_):c(c_),axis(0),pos(0){} void Set(int a,float p) const {const_cast<SpherePred*>(this)->axis=a;const_cast<SpherePred*>(this)->pos=p;} bool operator()(unsigned i) const { return c[3*i+axis] < pos; } };
int main(int argc,char**argv){ unsigned n = argc>1? atoi(argv[1]):1000000; std::mt19937 g(1); std::uniform_real_distribution<float> u(-1,1);
 std::vector<float> c(3*n), r(n); for(auto&x:c)x=u(g); for(auto&x:r)x=0.001f+0.002f*(u(g)+1);
 SphereGeom geom{c.data(), r.data()}; SpherePred pred(c.data());
 nanort::BVHAccel<float> a; nanort::BVHBuildOptions<float> o;
 auto t0=std::chrono::steady_clock::now(); bool ok=a.Build(n, geom, pred, o); auto t1=std::chrono::steady_clock::now();
 unsigned long long h=1469598103934665603ull; const unsigned char*p=(const unsigned char*)a.GetNodes().data(); size_t bytes=a.GetNodes().size()*sizeof(nanort::BVHNode<float>);
 // hash fields (leaf axis is defined here)
 for(size_t i=0;i<bytes;i++){h^=p[i];h*=1099511628211ull;} const unsigned char*q=(const unsigned char*)a.GetIndices().data(); for(size_t i=0;i<4*(size_t)n;i++){h^=q[i];h*=1099511628211ull;}
 printf("ok %d nodes %zu depth %u leaves %u branches %u hash %016llx  %.1f ms threads %u\n", ok, a.GetNodes().size(), a.GetStatistics().max_tree_depth, a.GetStatistics().num_leaf_nodes, a.GetStatistics().num_branch_nodes, h, std::chrono::duration<double,std::milli>(t1-t0).count(), nanort::detail::HostThreads()); }
