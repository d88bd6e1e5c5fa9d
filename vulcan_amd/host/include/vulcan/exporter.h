// exporter.h — upstream header name (ref: include/vulcan/exporter.h); the classes live in meshing.h
#pragma once
#include <vulcan/meshing.h>
