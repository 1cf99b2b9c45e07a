// Object.h — same record surface as the reference's Object.h (src/Object.h:8-93): Material 80 B, Sphere 32 B,
// Triangle 64 B, Node 48 B (byte-identical to the std430 structs of rt.comp:33-48 and to PtMaterial /
// PtSphere / PtTriangle / PtNode of include/pt_hip.h), Scene with materials/spheres/triangles, Mesh with
// vertices/indices/tris/nodes.  Instead of GL buffers, Scene::SetupSSBOs/UpdateSSBOs and the Mesh
// constructor hand their arrays to the HIP layer (pt_upload_scene / pt_upload_mesh), which lays them out as
// device SoA.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "Math.h"

struct Material {
    ptm::vec4 baseColor = ptm::vec4(1);
    ptm::vec4 specularColor = ptm::vec4(1);
    ptm::vec4 emissionColor = ptm::vec4(1);
    float smoothness = 0.0f;
    float specularSmoothness = 0.0f;
    float emissionStrength = 0.0f;
    float ior = 1.5f;
    float refractionAmount = 0.0f;
    float specularChance = 0.0f;
private:
    int pad[2] = {0, 0};
};

struct Sphere {
    ptm::vec3 position = ptm::vec3(0);
    float radius = 0.0f;
    unsigned int materialIndex = 0;

    Sphere(struct Scene &scene, ptm::vec3 pos, float rad, unsigned int materialIndex);   // pushes itself into the scene
private:
    int pad[3] = {0, 0, 0};
};

struct Triangle {
    ptm::vec4 p1 = ptm::vec4(0);
    ptm::vec4 p2 = ptm::vec4(0);
    ptm::vec4 p3 = ptm::vec4(0);
    unsigned int materialIndex = 0;

    Triangle() = default;
    Triangle(struct Scene &scene, ptm::vec3 _p1, ptm::vec3 _p2, ptm::vec3 _p3, unsigned int materialIndex);
    ptm::vec3 Center() const;
private:
    int pad[3] = {0, 0, 0};
};

struct Node {
    ptm::vec4 boundsMin = ptm::vec4(1e30f);
    ptm::vec4 boundsMax = ptm::vec4(-1e30f);
    int triIndex = 0;
    int numTris = 0;
    int childrenIndex = 0;

    void GrowBounds(const Triangle &tri);
private:
    int pad = 0;
};

static_assert(sizeof(Material) == 80 && sizeof(Sphere) == 32 && sizeof(Triangle) == 64 && sizeof(Node) == 48, "std430 layouts");

struct Scene {
    std::vector<Material> materials;
    std::vector<Sphere> spheres;
    std::vector<Triangle> triangles;

    void SetupSSBOs();      // src/Object.cpp:3-23  -> pt_upload_scene
    void UpdateSSBOs();     // src/Object.cpp:25-41 -> pt_upload_scene
};

struct Mesh {
    std::vector<ptm::vec4> vertices;
    std::vector<ptm::ivec4> indices;
    std::vector<Triangle> tris;
    uint32_t materialIndex = 0;
    std::vector<Node> nodes;

    Mesh(const char *filePath, uint32_t materialIndex);                     // load OBJ + root AABB + upload (src/Object.cpp:89-111)
    Mesh(std::vector<Triangle> triangles, uint32_t materialIndex);          // same from triangles already in memory
    // Several meshes as one (the reference's constructors rebind SSBO 1/2, so it only ever traces the mesh built last,
    // src/Object.cpp:104-110): triangles concatenated in the order given, each with its own materialIndex, one root
    // AABB over all of them.  Uploads the merged mesh.
    explicit Mesh(const std::vector<const Mesh *> &parts);
    bool ok() const { return loaded; }
private:
    bool loaded = false;
    void Load(const char *filePath);        // `v x y z` and `f a b c` records; also accepts `f a/b/c` tokens
    void GenBoundingBox();
    void Upload();
};
