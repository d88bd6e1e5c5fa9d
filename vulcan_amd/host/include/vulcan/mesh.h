// mesh.h — upstream header name (ref: include/vulcan/mesh.h); the classes live in meshing.h
#pragma once
#include <vulcan/meshing.h>
