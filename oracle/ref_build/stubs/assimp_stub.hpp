// STUB of the assimp types mesh_map's loader code names (never executed: the harness serves meshes from
// the in-memory map store).
#pragma once
#include <string>
struct aiString { std::string s; const char* C_Str() const { return s.c_str(); } };
struct aiVector3D { float x = 0, y = 0, z = 0; };
struct aiColor4D { float r = 0, g = 0, b = 0, a = 0; };
struct aiQuaternion { float w = 1, x = 0, y = 0, z = 0; aiVector3D Rotate(const aiVector3D& v) const { return v; } };
struct aiMatrix4x4
{
  aiMatrix4x4 operator*(const aiMatrix4x4&) const { return *this; }
  void Decompose(aiVector3D&, aiQuaternion&, aiVector3D&) const { }
};
inline aiVector3D operator*(const aiMatrix4x4&, const aiVector3D& v) { return v; }
struct aiFace { unsigned int mNumIndices = 0; unsigned int* mIndices = nullptr; };
struct aiMesh
{
  aiString mName; unsigned int mNumVertices = 0, mNumFaces = 0;
  aiVector3D* mVertices = nullptr; aiVector3D* mNormals = nullptr; aiFace* mFaces = nullptr; aiColor4D* mColors[8] = { };
  bool HasNormals() const { return mNormals != nullptr; }
  bool HasVertexColors(unsigned i) const { return mColors[i] != nullptr; }
};
struct aiNode
{
  aiString mName; aiMatrix4x4 mTransformation; unsigned int mNumChildren = 0; aiNode** mChildren = nullptr;
  unsigned int mNumMeshes = 0; unsigned int* mMeshes = nullptr;
};
struct aiScene { aiNode* mRootNode = nullptr; aiMesh** mMeshes = nullptr; };
enum { aiProcess_Triangulate = 1, aiProcess_JoinIdenticalVertices = 2, aiProcess_GenNormals = 4, aiProcess_ValidateDataStructure = 8, aiProcess_FindInvalidData = 16 };
#define AI_CONFIG_IMPORT_COLLADA_IGNORE_UP_DIRECTION "IMPORT_COLLADA_IGNORE_UP_DIRECTION"
namespace Assimp
{
class Importer
{
public:
  void SetPropertyBool(const char*, bool) { }
  const aiScene* ReadFile(const std::string&, unsigned) { return nullptr; }
  const char* GetErrorString() const { return "stub assimp: not available"; }
};
}
