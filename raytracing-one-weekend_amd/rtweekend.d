rtweekend: host/main.cpp host/render.h host/primitives.h \
  host/common-model.h host/vec3.h
host/render.h:
host/primitives.h:
host/common-model.h:
host/vec3.h:
