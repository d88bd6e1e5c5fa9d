// vulcan.h — umbrella header (ref: include/vulcan/vulcan.h.in)
#pragma once

#include <vulcan/block.h>
#include <vulcan/buffer.h>
#include <vulcan/color_integrator.h>
#include <vulcan/depth_integrator.h>
#include <vulcan/depth_tracker.h>
#include <vulcan/detector.h>
#include <vulcan/device.h>
#include <vulcan/exception.h>
#include <vulcan/frame.h>
#include <vulcan/hash.h>
#include <vulcan/image.h>
#include <vulcan/integrator.h>
#include <vulcan/light.h>
#include <vulcan/light_integrator.h>
#include <vulcan/math.h>
#include <vulcan/matrix.h>
#include <vulcan/projection.h>
#include <vulcan/pyramid_tracker.h>
#include <vulcan/tracer.h>
#include <vulcan/tracker.h>
#include <vulcan/transform.h>
#include <vulcan/types.h>
#include <vulcan/volume.h>
#include <vulcan/voxel.h>
