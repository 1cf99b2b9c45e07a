#include "Object.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "Renderer.h"

using namespace ptm;

static void upload_scene(Scene &s)
{
    Renderer::Check(pt_upload_scene(Renderer::ctx, reinterpret_cast<const PtMaterial *>(s.materials.data()), (int)s.materials.size(),
                                    reinterpret_cast<const PtSphere *>(s.spheres.data()), (int)s.spheres.size(),
                                    reinterpret_cast<const PtTriangle *>(s.triangles.data()), (int)s.triangles.size()),
                    "pt_upload_scene");
}

void Scene::SetupSSBOs() { upload_scene(*this); }
void Scene::UpdateSSBOs() { upload_scene(*this); }

Sphere::Sphere(Scene &scene, vec3 pos, float rad, unsigned int matIndex) : position(pos), radius(rad), materialIndex(matIndex)
{
    scene.spheres.push_back(*this);
}

Triangle::Triangle(Scene &scene, vec3 a, vec3 b, vec3 c, unsigned int matIndex) : materialIndex(matIndex)
{
    p1 = vec4(a.x, a.y, a.z, 0.0f);
    p2 = vec4(b.x, b.y, b.z, 0.0f);
    p3 = vec4(c.x, c.y, c.z, 0.0f);
    scene.triangles.push_back(*this);
}

vec3 Triangle::Center() const { return vec3((p1.x + p2.x + p3.x) / 3, (p1.y + p2.y + p3.y) / 3, (p1.z + p2.z + p3.z) / 3); }

void Node::GrowBounds(const Triangle &tri)
{
    for (const vec4 *p : {&tri.p1, &tri.p2, &tri.p3}) {
        boundsMin = min(boundsMin, *p);     // w takes part too, as in the reference: it ends at 0
        boundsMax = max(boundsMax, *p);
    }
}

Mesh::Mesh(const char *filePath, uint32_t matIndex) : materialIndex(matIndex)
{
    Load(filePath);
    GenBoundingBox();
    Upload();
}

Mesh::Mesh(std::vector<Triangle> triangles, uint32_t matIndex) : tris(std::move(triangles)), materialIndex(matIndex)
{
    for (Triangle &t : tris) t.materialIndex = matIndex;
    loaded = true;
    GenBoundingBox();
    Upload();
}

Mesh::Mesh(const std::vector<const Mesh *> &parts)
{
    loaded = !parts.empty();
    for (const Mesh *m : parts) {
        loaded = loaded && m->ok();
        tris.insert(tris.end(), m->tris.begin(), m->tris.end());
    }
    if (!parts.empty()) materialIndex = parts[0]->materialIndex;
    if (!loaded) { tris.clear(); return; }
    GenBoundingBox();
    Upload();
}

void Mesh::Load(const char *filePath)
{
    std::ifstream in(filePath);
    if (!in.is_open()) {
        std::cerr << "File not found: " << filePath << std::endl;
        return;
    }
    std::string line;
    while (std::getline(in, line)) {
        if (line.compare(0, 2, "v ") == 0) {
            vec4 v;
            if (std::sscanf(line.c_str(), "v %f %f %f", &v.x, &v.y, &v.z) == 3) vertices.push_back(v);
        } else if (line.compare(0, 2, "f ") == 0) {
            // "f a b c" as the reference reads it; "a/b/c" tokens keep only the vertex index
            int idx[3] = {0, 0, 0};
            const char *p = line.c_str() + 2;
            int got = 0;
            for (; got < 3; got++) {
                char *end = nullptr;
                long v = std::strtol(p, &end, 10);
                if (end == p) break;
                idx[got] = (int)v - 1;
                p = end;
                while (*p && *p != ' ' && *p != '\t') p++;      // skip "/vt/vn"
            }
            if (got == 3) { ivec4 f; f.x = idx[0]; f.y = idx[1]; f.z = idx[2]; indices.push_back(f); }
        }
    }
    for (const ivec4 &f : indices) {
        if (f.x < 0 || f.y < 0 || f.z < 0 || f.x >= (int)vertices.size() || f.y >= (int)vertices.size() || f.z >= (int)vertices.size()) {
            std::cerr << "face index out of range in " << filePath << std::endl;
            tris.clear();
            return;
        }
        Triangle t;
        t.p1 = vertices[f.x]; t.p2 = vertices[f.y]; t.p3 = vertices[f.z];
        t.materialIndex = materialIndex;
        tris.push_back(t);
    }
    loaded = true;
    std::cout << "'" << filePath << "': " << indices.size() << " triangles, " << vertices.size() << " vertices\n";
}

void Mesh::GenBoundingBox()
{
    Node root;
    for (const Triangle &t : tris) root.GrowBounds(t);
    root.numTris = (int)tris.size();
    root.childrenIndex = 1;
    nodes.assign(1, root);
}

void Mesh::Upload()
{
    // like the reference, the mesh constructed last is the one the kernel sees (it rebinds SSBO 1 and 2)
    Renderer::Check(pt_upload_mesh(Renderer::ctx, reinterpret_cast<const PtTriangle *>(tris.data()), (int)tris.size(),
                                   reinterpret_cast<const PtNode *>(nodes.data()), (int)nodes.size()),
                    "pt_upload_mesh");
}
