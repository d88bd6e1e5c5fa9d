// extractor.h — upstream header name (ref: include/vulcan/extractor.h); the classes live in meshing.h
#pragma once
#include <vulcan/meshing.h>
